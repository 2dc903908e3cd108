"""Drop-in import path of the reference (`from src.data import PairedMultimodalDataset, collate_coca_pair, ...`,
finetune_multimodal.py:22, finetune_text.py:22, finetune_image.py:13): re-exports the datasets and collate functions of
item_alignment_amd.data.datasets, which keep the reference's tuple layouts (src/data/data.py:37-240)."""
from item_alignment_amd.data.datasets import (RobertaOneTowerDataset, RobertaTwoTowerDataset, PKGMOneTowerDataset, PKGMTwoTowerDataset,  # noqa: F401
                                              RobertaImageOneTowerDataset, RobertaImageTwoTowerDataset, PairedImageDataset,
                                              PairedMultimodalDataset, collate_one_tower, collate_two_tower, collate_image,
                                              collate_multimodal, collate_multimodal_two_tower, collate_coca_pair)
