"""JointLoss — weighted sum of mapped loss modules (reference ``torchok/losses/base.py:7-113``).
Same constructor contract (weights all-or-none -> ValueError, optional normalisation), same
``forward(**outputs) -> (total, {tag: loss})`` and mapping errors."""
from typing import Any, Dict, List, Optional, Tuple

from torch import Tensor
from torch.nn import Module, ModuleList


class JointLoss(Module):
    def __init__(self, losses: List[Module], mappings: List[Dict[str, str]], tags: List[Optional[str]],
                 weights: List[Optional[float]], normalize_weights: bool = True):
        super().__init__()
        self.losses = ModuleList(losses)
        self.tag2loss = {tag: loss for tag, loss in zip(tags, self.losses) if tag is not None}
        self.tags = tags
        self.mappings = mappings
        num_specified = len([w for w in weights if w is not None])
        if num_specified > 0 and num_specified != len(losses):
            raise ValueError('Loss weights must be either specified for each loss function or '
                             'not specified for any loss function')
        self.weights = [1.] * len(self.losses) if num_specified == 0 else list(weights)
        if normalize_weights:
            total = sum(self.weights)
            self.weights = [w / total for w in self.weights]

    def forward(self, **kwargs) -> Tuple[Tensor, Dict[str, Tensor]]:
        total_loss = None
        tagged = {}
        for loss_module, mapping, tag, weight in zip(self.losses, self.mappings, self.tags, self.weights):
            args = self._parse_match_csv(mapping, **kwargs)
            if not getattr(loss_module, 'consumes_lazy_logits', False):
                # SegmentationHead's lazy upsampling (losses/cross_entropy.py: UpsampledLogits) is understood by CrossEntropyLoss
                # only; every other loss module (ours or a user's) gets the real tensor with its autograd edge
                args = {k: (v.materialize() if hasattr(v, 'materialize') and hasattr(v, '_low') else v) for k, v in args.items()}
            loss = loss_module(**args)
            # `0. + loss * weight` of the reference; x*1.0 and 0.+x are exact, so skip those launches
            term = loss if weight == 1.0 else loss * weight
            total_loss = term if total_loss is None else total_loss + term
            if tag is not None:
                tagged[tag] = loss
        if total_loss is None:
            total_loss = 0.
        return total_loss, tagged

    def __getitem__(self, tag: str) -> Module:
        if tag in self.tag2loss:
            return self.tag2loss[tag]
        raise KeyError(f'Cannot access loss {tag}. You should tag your losses for direct access with a tag key')

    @staticmethod
    def _parse_match_csv(mapping: Dict[str, str], **model_outputs) -> Dict[str, Any]:
        out = {}
        for target_arg, source_arg in mapping.items():
            if source_arg in model_outputs:
                out[target_arg] = model_outputs[source_arg]
            else:
                raise ValueError(f'Cannot find {source_arg} for your mapping {target_arg} : {source_arg}. '
                                 f'You should either add {source_arg} output to your model or remove the mapping '
                                 f'from configuration')
        return out
