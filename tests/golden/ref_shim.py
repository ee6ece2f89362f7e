"""Import shim for the *reference* GOAT model code (TEST TOOLING, this container only).

Used only by `tests/golden/make_golden_*.py` to generate golden vectors from the reference's own
Python (mounted read-only at /root/reference).  Nothing here is shipped to, or imported on, the GPU
box: `/root/reference` does not exist there, and the `-m gpu` tests / smoke / bench never import
this module.

What it does:
  1. stubs `pynvml` (pulled in by pretrain_src/data/common.py:6);
  2. restores `transformers.modeling_utils.apply_chunking_to_forward` (moved in transformers 5.x;
     pretrain_src/model/Bert_backbone.py:10-13);
  3. swaps `transformers.BertPreTrainedModel` for a small nn.Module stand-in with the four
     behaviours of transformers 4.34.1 the reference relies on (config storage, `_init_weights`,
     `init_weights`, `_tie_or_clone_weights`).
The pretrain tree and the fine-tune tree must be imported in separate processes (both define a
top-level `utils` package).
"""
import sys
import types

import torch
from torch import nn

REF_ROOT = '/root/reference'


def _install_common():
    if 'pynvml' not in sys.modules:
        sys.modules['pynvml'] = types.ModuleType('pynvml')
    import transformers
    import transformers.models.bert.modeling_bert  # noqa: F401  (forces lazy-module resolution)
    import transformers.models.roberta.modeling_roberta  # noqa: F401
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    if not hasattr(mu, 'apply_chunking_to_forward'):
        mu.apply_chunking_to_forward = pu.apply_chunking_to_forward

    class BertPreTrainedModelStandIn(nn.Module):
        """transformers==4.34.1 BertPreTrainedModel behaviours used by the reference."""

        def __init__(self, config, *a, **k):
            super().__init__()
            self.config = config

        def _init_weights(self, module):
            if isinstance(module, nn.Linear):
                module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
                if module.bias is not None:
                    module.bias.data.zero_()
            elif isinstance(module, nn.Embedding):
                module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
                if module.padding_idx is not None:
                    module.weight.data[module.padding_idx].zero_()
            elif isinstance(module, nn.LayerNorm):
                module.bias.data.zero_()
                module.weight.data.fill_(1.0)

        def init_weights(self):
            self.apply(self._init_weights)
            self.tie_weights()

        def tie_weights(self):
            pass

        def _tie_or_clone_weights(self, output_embeddings, input_embeddings):
            output_embeddings.weight = input_embeddings.weight

    transformers.BertPreTrainedModel = BertPreTrainedModelStandIn
    return transformers


def make_config(json_path, **overrides):
    """PretrainedConfig + the 4.34 defaults transformers 5.x dropped + reference-injected attrs."""
    from transformers import PretrainedConfig
    cfg = PretrainedConfig.from_json_file(json_path)
    for k, v in dict(pad_token_id=None, is_decoder=False, add_cross_attention=False,
                     chunk_size_feed_forward=0).items():
        if not hasattr(cfg, k) or getattr(cfg, k) is None:
            setattr(cfg, k, v)
    cfg.empty_cache = False
    cfg.cuda_first_device = 0
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def import_pretrain():
    """Returns the reference module `model.pretrain_goat` (pretrain_src tree)."""
    _install_common()
    p = REF_ROOT + '/pretrain_src'
    if p not in sys.path:
        sys.path.insert(0, p)
    import model.pretrain_goat as pg  # noqa
    return pg


def import_nav():
    """Returns the reference module `models.vilmodel_GOAT` (map_nav_src tree)."""
    _install_common()
    p = REF_ROOT + '/map_nav_src'
    if p not in sys.path:
        sys.path.insert(0, p)
    import models.vilmodel_GOAT as vg  # noqa
    return vg
