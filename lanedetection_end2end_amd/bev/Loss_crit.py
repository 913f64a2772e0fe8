"""``Loss_crit`` of the BEV tree (BEV/Loss_crit.py): same public names."""
from ..losses import Area_Loss, CrossEntropyLoss2d, MSE_Loss, polynomial  # noqa: F401
from ..losses import define_loss_crit_bev as define_loss_crit  # noqa: F401
