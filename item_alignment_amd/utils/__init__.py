"""Constants and logger of the reference's src/utils (config.py:1-6, logger.py:1-11)."""
import logging

ROBERTA_WEIGHTS_NAME = "pytorch_model.bin"
KG_WEIGHTS_NAME = "pkgm_model.bin"
COCA_WEIGHTS_NAME = "coca_model.bin"
VIT_WEIGHTS_NAME = "image_encoder.bin"
BOS_TOKEN = "<S>"

logging.basicConfig(format="%(asctime)s %(levelname)-4s [%(filename)s:%(lineno)s]  %(message)s", datefmt="%Y/%m/%d %H:%M:%S",
                    level=logging.INFO)
logger = logging.getLogger("item_alignment_amd")
