#!/usr/bin/env python
"""Retrieval-meter micro-benchmark at validation scale (Stanford Online Products test split: 60 502 images, the
`pairwise_sop.yaml` / `triplet_sop.yaml` recipes; embedding 512):  python tools/ubench/retrieval_bench.py [--n 60502] [--d 512]
Prints the time of the exhaustive search kernels and of a whole HitAtKMeter.compute()."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import torchok_amd as T  # noqa: E402
from torchok_amd import _C  # noqa: E402
from torchok_amd.engine.core import ptr, stream_ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=60502)
    ap.add_argument('--d', type=int, default=512)
    ap.add_argument('--k', type=int, default=1)
    a = ap.parse_args()
    g = torch.Generator(device='cuda').manual_seed(0)
    classes = a.n // 5
    labels = torch.arange(a.n, device='cuda') % classes
    centers = torch.randn(classes, a.d, generator=g, device='cuda')
    vec = centers[labels] + 2.0 * torch.randn(a.n, a.d, generator=g, device='cuda')
    lib, st = _C.lib(), stream_ptr()
    m = min(a.n, (1 << 30) // (4 * a.n))
    sim = torch.empty(m, a.n, device='cuda')
    vals = torch.empty(m, a.k + 1, device='cuda')
    idx = torch.empty(m, a.k + 1, dtype=torch.int64, device='cuda')

    def timed(fn, it=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    t = timed(lambda: _C.check(lib.tok_sim_matrix(ptr(vec), ptr(vec), m, a.n, a.d, a.d, a.d, 0, ptr(sim), a.n, st), 'sim'))
    print(f'tok_sim_matrix  {m} x {a.n} x {a.d}: {t:8.2f} ms  {2 * m * a.n * a.d / t / 1e9:7.1f} TFLOP/s fp32')
    t = timed(lambda: _C.check(lib.tok_topk_rows(ptr(sim), m, a.n, a.n, a.k + 1, ptr(vals), ptr(idx), st), 'topk'))
    print(f'tok_topk_rows   {m} rows x {a.n}, k={a.k + 1}: {t:8.2f} ms  {(a.k + 1) * m * a.n * 4 / t / 1e6:7.1f} GB/s read')
    meter = T.METRICS.get('HitAtKMeter')(dataset_type='classification', k=a.k, normalize_vectors=True)
    for lo in range(0, a.n, 4096):
        meter.update(vectors=vec[lo:lo + 4096], group_labels=labels[lo:lo + 4096])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = meter.compute()
    torch.cuda.synchronize()
    print(f'HitAtKMeter.compute() n={a.n} d={a.d} k={a.k}: {1e3 * (time.perf_counter() - t0):8.1f} ms  value {v:.4f}')


if __name__ == '__main__':
    main()
