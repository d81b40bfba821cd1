#!/usr/bin/env python
"""OCRSegmentationHead alone at the HRNet-W48 segmentation size (neck output 720 x 128 x 256 of a 512 x 1024 image, 19
classes): forward + backward time per call.  python tools/ubench/ocr_bench.py [--batch 8] [--mid 128] [--key 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import torchok_amd as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--mid', type=int, default=128)
    ap.add_argument('--key', type=int, default=64)
    a = ap.parse_args()
    head = T.HEADS.get('OCRSegmentationHead')(in_channels=720, num_classes=19, ocr_mid_channels=a.mid,
                                              ocr_key_channels=a.key).cuda().train()
    feats = torch.randn(a.batch, 720, 128, 256, device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    feats.requires_grad_(True)
    image = torch.zeros(a.batch, 3, 512, 1024, device='cuda')

    def step():
        out, aux = head([image, feats])
        (out.float().mean() + 0.4 * aux.float().mean()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f'OCRSegmentationHead(720 -> {a.mid}/{a.key}, 19 classes) B={a.batch} 128x256 -> 512x1024: '
          f'{e0.elapsed_time(e1) / 5:.2f} ms per forward + backward')


if __name__ == '__main__':
    main()
