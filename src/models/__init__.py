"""Drop-in import path of the reference (`from src.models import RobertaOneTower, ...`): re-exports the
MI355X-native implementations from item_alignment_amd.models."""
from item_alignment_amd.models import *  # noqa: F401,F403
from item_alignment_amd.models import (RobertaModel, RobertaOneTower, RobertaTwoTower, PKGMOneTower, PKGMTwoTower, TextCNNTwoTower,  # noqa: F401
                                       RobertaImageOneTower, RobertaImageTwoTower, CoCaForItemAlignment, NFNetTwoTower, VitTwoTower,
                                       ResNetTwoTower)
