"""The reference's own train scripts import their models, data formats and utilities through `src.*`
(finetune_multimodal.py:16-30, finetune_text.py:15-30, finetune_image.py:12-14, model_soup_multimodal.py:15-31): the same
import lists must resolve against this repository's shims ("keep the reference's train script")."""


def test_reference_import_lists_resolve():
    from src.models import RobertaModel, RobertaImageOneTower, RobertaImageTwoTower, CoCaForItemAlignment  # noqa: F401
    from src.models import PKGMOneTower, PKGMTwoTower, RobertaTwoTower, RobertaOneTower, TextCNNTwoTower  # noqa: F401
    from src.models import NFNetTwoTower, VitTwoTower, ResNetTwoTower  # noqa: F401
    from src.data import (RobertaImageOneTowerDataset, RobertaImageTwoTowerDataset, collate_multimodal, PairedMultimodalDataset,  # noqa: F401
                          collate_coca_pair, collate_multimodal_two_tower)
    from src.data import (PKGMTwoTowerDataset, PKGMOneTowerDataset, RobertaOneTowerDataset, RobertaTwoTowerDataset,  # noqa: F401
                          collate_one_tower, collate_two_tower)
    from src.data import PairedImageDataset, collate_image  # noqa: F401
    from src.utils import logger, VIT_WEIGHTS_NAME, ROBERTA_WEIGHTS_NAME, BOS_TOKEN  # noqa: F401
    assert callable(collate_coca_pair) and callable(collate_image)


def test_config_directory_of_the_reference_layout_exists():
    """INTEGRATION.md points --config_file at src/config/<name>.json like the reference's scripts do."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("roberta_large.json", "coca_large.json", "pkgm_large.json", "eca_nfnet_l0.json", "resnetv2_50.json"):
        assert os.path.exists(os.path.join(root, "src", "config", name)), name
