"""Operator API of the reference's src/models package (src/models/__init__.py:1-6 star-imports), on the HIP engine."""
from .loss import EuclideanDistanceLoss, HingeLoss
from .base import (InnerProduct, VecSimClassificationHead, TwoTowerClassificationHead, RobertaClassificationHead,
                   SequenceClassifierOutput, create_position_ids_from_input_ids, RobertaEmbeddings, RobertaPKGMEmbeddings,
                   RobertaImageEmbeddings, RobertaEncoder)
from .text import (RobertaModel, RobertaOneTower, RobertaTwoTower, RobertaPKGMModel, PKGMOneTower, PKGMTwoTower, TextCNN,
                   TextCNNTwoTower)
from .image import VisionTransformer, VitTwoTower, NFNetTwoTower, ResNetTwoTower, create_model
from .multimodal import (RobertaImageModel, RobertaImageOneTower, RobertaImageTwoTower, CoCaModel, CoCaForItemAlignment, LayerNorm,
                         Residual, RotaryEmbedding, SwiGLU, ParallelTransformerBlock, CrossAttention)
