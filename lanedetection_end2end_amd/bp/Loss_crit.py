"""``Loss_crit`` of the BP tree (BP/Loss_crit.py): same public names."""
from ..losses import Area_Loss, CrossEntropyLoss2d, MSE_Loss, backprojection_loss, polynomial  # noqa: F401
from ..losses import define_loss_crit_bp as define_loss_crit  # noqa: F401
