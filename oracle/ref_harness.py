"""TEST INFRASTRUCTURE — container-only harness that imports the reference's own Python model code
from /root/reference (read-only) so golden vectors can be captured from it.  Nothing here ships or
travels to the GPU box; only tests/golden/*.npz (data) does.

The reference pins transformers==4.20.1 / timm==0.6.5; this container has transformers 5.x and no
timm, so three plumbing shims (no arithmetic) are applied, exactly as SURVEY.md Appendix B lists:
  S1  RobertaPreTrainedModel.get_head_mask (removed in 5.x; reference text.py:1231, multimodal.py:164)
  S2  SequenceClassifierOutput as a dataclass with a `logits` field (reference base.py:160-186)
  S3  RobertaEncoder returns hidden_states (reference reads outputs.hidden_states, text.py:1452)
and stub modules stand in for timm / torch_geometric / jieba (only their names are needed at import).
"""
import dataclasses
import sys
import types
from typing import Optional

REFERENCE_ROOT = "/root/reference"


def load_reference():
    """Returns the reference's `src.models` package (with shims applied)."""
    import torch
    import transformers  # noqa: F401  (must be imported before the stubs are installed)
    from transformers import RobertaPreTrainedModel
    from transformers.utils import ModelOutput
    from transformers.models.roberta import modeling_roberta as MR

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    class _Stub(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return type(n, (), {})

    for m in ["timm", "timm.models", "timm.models.nfnet", "timm.models.vision_transformer", "timm.models.layers",
              "timm.models.layers.classifier", "timm.data", "timm.data.transforms_factory", "torch_geometric",
              "torch_geometric.nn", "jieba"]:
        if m not in sys.modules:
            s = _Stub(m)
            s.__path__ = []
            sys.modules[m] = s

    import src.models as M
    import src.models.base as B
    import src.models.text as T
    import src.models.multimodal as MM
    import src.models.image as IM

    RobertaPreTrainedModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n  # S1

    @dataclasses.dataclass
    class SCO(ModelOutput):  # S2
        loss: Optional[torch.Tensor] = None
        logits: Optional[torch.Tensor] = None
        probs: Optional[torch.Tensor] = None
        src_embeds: Optional[torch.Tensor] = None
        tgt_embeds: Optional[torch.Tensor] = None

    for mod in (B, T, MM, M, IM):
        mod.SequenceClassifierOutput = SCO

    # S3: collect hidden states with forward hooks on the layers
    if not getattr(MR.RobertaEncoder, "_ia_patched", False):
        orig_forward = MR.RobertaEncoder.forward

        def forward(self, hidden_states, *args, **kwargs):
            collected = [hidden_states]
            hooks = [layer.register_forward_hook(lambda mod, inp, out: collected.append(out[0] if isinstance(out, tuple) else out))
                     for layer in self.layer]
            try:
                out = orig_forward(self, hidden_states, *args, **kwargs)
            finally:
                for h in hooks:
                    h.remove()
            try:
                out.hidden_states = tuple(collected)
            except Exception:
                pass
            return out

        MR.RobertaEncoder.forward = forward
        MR.RobertaEncoder._ia_patched = True
    return M


def reference_config(json_path=None, **overrides):
    """BertConfig the way finetune_text.py:195-210 / finetune_multimodal.py:185-199 build it."""
    from transformers import BertConfig
    cfg = BertConfig.from_json_file(json_path) if json_path else BertConfig()
    defaults = dict(interaction_type="one_tower", classification_method="cls", similarity_measure="NA", loss_type="ce",
                    max_seq_len=None, max_seq_len_pv=None, max_pvs=0, loss_margin=1.0, cls_layers="1", cls_pool="cat",
                    ensemble=None, auxiliary_task=False, image_hidden_size=3072, image_size=384, filter_sizes="1,2,3,5",
                    num_filters=36, classifier_dropout=None)
    for k, v in {**defaults, **overrides}.items():
        setattr(cfg, k, v)
    cfg._attn_implementation = "eager"
    return cfg
