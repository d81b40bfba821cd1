"""OCRSegmentationHead (reference ``torchok/models/heads/segmentation/ocr.py:134-192``: HRNet + object-contextual
representations) with the reference's module tree (``conv3x3_ocr``, ``ocr_gather_head``, ``ocr_distri_head.
object_context_block.{f_pixel,f_object,f_down,f_up}``, ``ocr_distri_head.conv_bn_dropout``, ``last_reduction``,
``aux_head``, ``classifier``), so its checkpoints load.  The whole head is one engine region: the ConvBnRelu units run on
the conv kernels (the per-class "proxy" map (B, C, K, 1) is a (B, K, 1, C) NHWC tensor for them), SpatialGather and the
object attention on the pixel-by-class kernels of ``engine/ocr.py``.

As in the reference, ``forward`` returns ``(out, out_aux)`` in training mode and ``out`` in eval mode (:189-192)."""
from typing import List, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import HEADS
from ...engine import functional as EF
from ...engine import ocr as EO
from ...engine import resample as ER
from ..base import BaseModel
from ..modules.convbnact import ConvBnAct


class SpatialGather_Module(nn.Module):
    def __init__(self, num_classes: int = 0, scale: int = 1):
        super().__init__()
        self.num_classes, self.scale = num_classes, scale

    def run(self, r, feats, probs):
        return EO.spatial_gather(r, feats, probs, float(self.scale))


class ObjectAttentionBlock(nn.Module):
    def __init__(self, in_channels: int, key_channels: int, scale: int = 1):
        super().__init__()
        if scale != 1:
            raise NotImplementedError('torchok_amd ObjectAttentionBlock: scale = 1 (what OCRSegmentationHead uses)')
        self.scale, self.in_channels, self.key_channels = scale, in_channels, key_channels
        self.pool = nn.MaxPool2d(kernel_size=(scale, scale))
        two = lambda: nn.Sequential(ConvBnAct(in_channels, key_channels, kernel_size=1),       # noqa: E731
                                    ConvBnAct(key_channels, key_channels, kernel_size=1))
        self.f_pixel, self.f_object, self.f_down = two(), two(), two()
        self.f_up = ConvBnAct(key_channels, in_channels, kernel_size=1)

    @staticmethod
    def _seq(r, seq, x):
        for m in seq:
            x = m.run(r, x)
        return x

    def run(self, r, x, proxy):
        query = self._seq(r, self.f_pixel, x)
        key = self._seq(r, self.f_object, proxy)
        value = self._seq(r, self.f_down, proxy)
        context = EO.object_attention(r, query, key, value, self.key_channels ** -.5)
        return self.f_up.run(r, context)


class SpatialOCR(nn.Module):
    def __init__(self, in_channels: int, key_channels: int, out_channels: int, scale: int = 1, dropout: float = 0.1):
        super().__init__()
        self.object_context_block = ObjectAttentionBlock(in_channels, key_channels, scale)
        self.conv_bn_dropout = nn.Sequential(ConvBnAct(2 * in_channels, out_channels, kernel_size=1), nn.Dropout2d(dropout))

    def draw_dropout(self, batch: int, channels: int, device):
        """Keep/scale factors of nn.Dropout2d: one Bernoulli draw per (image, channel) (F.dropout2d -> feature_dropout)."""
        drop = self.conv_bn_dropout[1]
        if not drop.training or drop.p == 0.:
            return None
        keep = 1. - drop.p
        return torch.empty((batch, channels), dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def run(self, r, feats, proxy):
        context = self.object_context_block.run(r, feats, proxy)
        h, w = feats.shape[1:3]
        cat = ER.bilinear_concat(r, [context, feats], (h, w))          # same size: a channel concatenation
        out = self.conv_bn_dropout[0].run(r, cat)
        return EO.channel_dropout(r, out, self.draw_dropout(out.shape[0], out.c, out.data.device))


@HEADS.register_class
class OCRSegmentationHead(BaseModel):
    def __init__(self, in_channels: int, num_classes: int, do_interpolate: bool = True, ocr_mid_channels=128,
                 ocr_key_channels=64):
        super().__init__(in_channels, num_classes)
        self.do_interpolate, self.num_classes = do_interpolate, num_classes
        self.conv3x3_ocr = ConvBnAct(in_channels, ocr_mid_channels, kernel_size=3, padding=1)
        self.ocr_gather_head = SpatialGather_Module(num_classes)
        self.ocr_distri_head = SpatialOCR(in_channels=ocr_mid_channels, key_channels=ocr_key_channels,
                                          out_channels=ocr_mid_channels, scale=1, dropout=0.05)
        self.last_reduction = ConvBnAct(ocr_mid_channels, ocr_mid_channels // 16, kernel_size=1, stride=1, padding=0)
        self.aux_head = nn.Sequential(ConvBnAct(in_channels, in_channels, kernel_size=1, stride=1, padding=0),
                                      nn.Conv2d(in_channels, num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        self.classifier = nn.Conv2d(ocr_mid_channels // 16, num_classes, kernel_size=1)

    def forward(self, feats: List[Tensor]) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        input_image, feats = feats
        size = tuple(input_image.shape[2:])
        with engine.region() as r:
            x = r.input(feats)
            out_aux = EF.conv_bn_act(r, self.aux_head[0].run(r, x), self.aux_head[1], None, False, None)
            f = self.conv3x3_ocr.run(r, x)
            context = self.ocr_gather_head.run(r, f, out_aux)
            f = self.ocr_distri_head.run(r, f, context)
            f = self.last_reduction.run(r, f)
            out = EF.conv_bn_act(r, f, self.classifier, None, False, None)
            if self.do_interpolate:
                out = ER.bilinear_resize(r, out, size)
                aux_up = ER.bilinear_resize(r, out_aux, size)
            else:
                aux_up = out_aux
            if self.training:
                out, aux_up = r.output(out, aux_up)
            else:
                out, aux_up = r.output(out), None
        if self.num_classes == 1:
            out = out[:, 0]
            aux_up = aux_up[:, 0] if aux_up is not None else None
        return (out, aux_up) if self.training else out
