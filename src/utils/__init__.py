from item_alignment_amd.utils import *  # noqa: F401,F403
from item_alignment_amd.utils import logger, ROBERTA_WEIGHTS_NAME, KG_WEIGHTS_NAME, COCA_WEIGHTS_NAME, VIT_WEIGHTS_NAME, BOS_TOKEN  # noqa: F401
