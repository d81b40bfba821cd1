"""CPU restatement of the reference's OCRSegmentationHead (second half): plain PyTorch fp32, each piece
citing torchok/models/heads/segmentation/ocr.py.  TEST INFRASTRUCTURE ONLY.

Pinned by tests/golden/ocr_head_step.npz: tests/golden/gen_golden.py imports the reference's OWN ocr.py and convbnact.py
(both in-tree, no third-party imports beyond torch), runs forward + backward in training mode and asserts this restatement
is bit-identical (both outputs, every gradient).  Parameter names equal the reference's."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvBnRelu(nn.Module):
    """modules/bricks/convbnact.py:9-62 with act_layer = ReLU."""

    def __init__(self, cin, cout, kernel_size, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


def spatial_gather(feats, probs, scale=1):
    """:37-46."""
    b, k = probs.shape[:2]
    p = F.softmax(scale * probs.view(b, k, -1), dim=2)
    f = feats.view(b, feats.size(1), -1).permute(0, 2, 1)
    return torch.matmul(p, f).permute(0, 2, 1).unsqueeze(3)


class ObjectAttentionBlock(nn.Module):
    """:49-104 (scale = 1)."""

    def __init__(self, cin, ck):
        super().__init__()
        self.ck = ck
        self.pool = nn.MaxPool2d(kernel_size=(1, 1))
        two = lambda: nn.Sequential(ConvBnRelu(cin, ck, 1), ConvBnRelu(ck, ck, 1))      # noqa: E731
        self.f_pixel, self.f_object, self.f_down = two(), two(), two()
        self.f_up = ConvBnRelu(ck, cin, 1)

    def forward(self, x, proxy):
        b, _, h, w = x.shape
        q = self.f_pixel(x).view(b, self.ck, -1).permute(0, 2, 1)
        k = self.f_object(proxy).view(b, self.ck, -1)
        v = self.f_down(proxy).view(b, self.ck, -1).permute(0, 2, 1)
        sim = F.softmax((self.ck ** -.5) * torch.matmul(q, k), dim=-1)
        ctx = torch.matmul(sim, v).permute(0, 2, 1).contiguous().view(b, self.ck, h, w)
        return self.f_up(ctx)


class SpatialOCR(nn.Module):
    """:107-131."""

    def __init__(self, cin, ck, cout, dropout):
        super().__init__()
        self.object_context_block = ObjectAttentionBlock(cin, ck)
        self.conv_bn_dropout = nn.Sequential(ConvBnRelu(2 * cin, cout, 1), nn.Dropout2d(dropout))

    def forward(self, feats, proxy):
        return self.conv_bn_dropout(torch.cat([self.object_context_block(feats, proxy), feats], 1))


class OCRSegmentationHead(nn.Module):
    """:134-192."""

    def __init__(self, in_channels, num_classes, do_interpolate=True, ocr_mid_channels=128, ocr_key_channels=64):
        super().__init__()
        self.do_interpolate, self.num_classes = do_interpolate, num_classes
        self.conv3x3_ocr = ConvBnRelu(in_channels, ocr_mid_channels, 3, padding=1)
        self.ocr_gather_head = nn.Module()
        self.ocr_distri_head = SpatialOCR(ocr_mid_channels, ocr_key_channels, ocr_mid_channels, 0.05)
        self.last_reduction = ConvBnRelu(ocr_mid_channels, ocr_mid_channels // 16, 1)
        self.aux_head = nn.Sequential(ConvBnRelu(in_channels, in_channels, 1), nn.Conv2d(in_channels, num_classes, 1))
        self.classifier = nn.Conv2d(ocr_mid_channels // 16, num_classes, 1)

    def forward(self, inputs):
        image, feats = inputs
        out_aux = self.aux_head(feats)
        feats = self.conv3x3_ocr(feats)
        context = spatial_gather(feats, out_aux)
        feats = self.ocr_distri_head(feats, context)
        out = self.classifier(self.last_reduction(feats))
        if self.do_interpolate:
            out = F.interpolate(out, image.shape[2:], mode='bilinear', align_corners=False)
            out_aux = F.interpolate(out_aux, image.shape[2:], mode='bilinear', align_corners=False)
        if self.num_classes == 1:
            out, out_aux = out[:, 0], out_aux[:, 0]
        return (out, out_aux) if self.training else out
