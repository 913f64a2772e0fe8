"""``Loss_crit`` of the BP tree (BP/Loss_crit.py): same public names."""
from lanedetection_end2end_amd.losses import Area_Loss, CrossEntropyLoss2d, MSE_Loss, backprojection_loss, polynomial  # noqa: F401
from lanedetection_end2end_amd.losses import define_loss_crit_bp as define_loss_crit  # noqa: F401
