"""``test`` of the BP tree (BP/test.py): the pieces of ``test_model`` that run on the device."""
from lanedetection_end2end_amd.clas import Projections, horizon_row, line_flags, resize_coordinates  # noqa: F401
