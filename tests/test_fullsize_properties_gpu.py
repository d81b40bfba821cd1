"""Size-independent properties at BASELINE.json's full size (ResNet-50, 224x224, batch 256, bf16) — where the oracle
cannot run in seconds, the HIP path is checked against itself through properties the domain offers:
  * linearity of the convolution kernels in a power-of-two scale (exact in bf16),
  * BatchNorm normalisation invariants of the fused statistics / finalize / apply chain,
  * additivity of the mean-loss gradient over a batch split with frozen (eval-mode) statistics,
  * bit-reproducibility of a full training step (deterministic reductions, two-stream schedule)."""
import ctypes

import pytest
import torch

import torchok_amd as T
from helpers import cls_config, deterministic_state
from torchok_amd import _C

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _task(seed=3, classes=1000):
    cfg = cls_config('resnet50', classes, inputs_shape=(3, 224, 224))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, seed)
    task.load_state_dict(sd, strict=False)
    return task.cuda()


def _batch(n=256, seed=1):
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(n, 3, 224, 224, generator=g, device='cuda').to(BF16)
    y = torch.randint(0, 1000, (n,), generator=g, device='cuda')
    return x, y


@pytest.mark.parametrize('shape', [(256, 56, 56, 64, 64, 3, 1, 1), (256, 56, 56, 256, 64, 1, 1, 0),
                                   (256, 14, 14, 1024, 2048, 1, 2, 0), (256, 7, 7, 512, 512, 3, 1, 1)])
def test_conv_kernels_are_linear_in_a_power_of_two_scale(shape):
    n, h, w, c, k, r, stride, pad = shape
    lib = _C.load_library()
    p = (h + 2 * pad - r) // stride + 1
    d = _C.ConvDesc(n, h, w, c, k, r, r, p, p, stride, pad, r)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(n, h, w, c, generator=g, device='cuda').to(BF16)
    wt = (torch.randn(k, r, r, c, generator=g, device='cuda') * 0.05)
    wf = torch.empty(k, r, r, c, dtype=BF16, device='cuda')
    wd = torch.empty(c, r, r, k, dtype=BF16, device='cuda')
    assert lib.tok_pack_weight_both(wt.data_ptr(), k, r, r, c, wf.data_ptr(), k, r, c, wd.data_ptr(), st) == 0
    y1, y4 = torch.empty(n, p, p, k, dtype=BF16, device='cuda'), torch.empty(n, p, p, k, dtype=BF16, device='cuda')
    x4 = x * 4
    assert lib.tok_conv_fwd(ctypes.byref(d), x.data_ptr(), wf.data_ptr(), None, y1.data_ptr(), None, st) == 0
    assert lib.tok_conv_fwd(ctypes.byref(d), x4.data_ptr(), wf.data_ptr(), None, y4.data_ptr(), None, st) == 0
    assert torch.equal(y4, y1 * 4)                               # forward: exact
    dx1, dx4 = torch.empty_like(x), torch.empty_like(x)
    dy = torch.randn(n, p, p, k, generator=g, device='cuda').to(BF16)
    dy4 = dy * 4
    assert lib.tok_conv_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx1.data_ptr(), 0, st) == 0
    assert lib.tok_conv_dgrad(ctypes.byref(d), dy4.data_ptr(), wd.data_ptr(), dx4.data_ptr(), 0, st) == 0
    assert torch.equal(dx4, dx1 * 4)                             # data gradient: exact
    wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.empty(max(wsb // 4, 16), device='cuda')
    dw1, dw4, dw1b = (torch.empty(k, r, r, c, device='cuda') for _ in range(3))
    for src, dst in ((dy, dw1), (dy4, dw4), (dy, dw1b)):
        assert lib.tok_conv_wgrad(ctypes.byref(d), x.data_ptr(), src.data_ptr(), dst.data_ptr(), k, c, ws.data_ptr(), wsb, 0,
                                  st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw4, dw1 * 4) and torch.equal(dw1, dw1b)  # weight gradient: exact, and reproducible


def test_batchnorm_chain_normalises_at_full_size():
    """conv -> fused statistics -> finalize -> apply (no ReLU, unit gamma, zero beta) gives per-channel mean 0 / var 1
    over the 802816 pixels of a 56x56x256 batch."""
    import torch.nn as nn
    from torchok_amd import engine
    from torchok_amd.engine import functional as EF
    conv = nn.Conv2d(64, 256, 1, bias=False).cuda()
    bn = nn.BatchNorm2d(256).cuda().train()
    x = (torch.randn(256, 64, 56, 56, device='cuda') * 3 + 1).to(BF16).to(memory_format=torch.channels_last)
    with torch.no_grad(), engine.region() as r:
        y = r.output(EF.conv_bn_act(r, r.input(x), conv, bn, relu=False))
    y = y.float()
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    assert float(mean.abs().max()) < 2e-3 and float((var - 1).abs().max()) < 1e-2
    assert int(bn.num_batches_tracked) == 1


def test_mean_loss_gradient_is_additive_over_a_batch_split():
    """Frozen statistics (eval-mode BatchNorm, frozen BN affine): grad(batch of 256) == mean of the gradients of its two
    halves, for every conv / fc weight (wgrad + dgrad + CE at full size; fp32 split-reduction noise only)."""
    task = _task().eval()
    for m in task.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.requires_grad_(False)
            m.bias.requires_grad_(False)
    x, y = _batch(256)

    def grads(xs, ys):
        for p in task.parameters():
            p.grad = None
        out = task.forward_with_gt({'image': xs, 'target': ys})
        task.losses(**out)[0].backward()
        return {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.requires_grad}
    full, a, b = grads(x, y), grads(x[:128], y[:128]), grads(x[128:], y[128:])
    worst = 0.0
    for n in full:
        ref = 0.5 * (a[n] + b[n])
        err = float((full[n] - ref).norm() / (ref.norm() + 1e-12))
        worst = max(worst, err)
    assert worst < 5e-3, worst


def test_full_size_training_step_is_bit_reproducible():
    res = []
    for _ in range(2):
        task = _task().train()
        opt = task.configure_optimizers()[0]['optimizer']
        x, y = _batch(256)
        for it in range(2):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad(set_to_none=True)
            out['loss'].backward()
            opt.step()
        torch.cuda.synchronize()
        assert torch.isfinite(out['loss'])
        res.append(torch.cat([p.detach().reshape(-1) for p in task.parameters()]))
    assert torch.equal(res[0], res[1])


# ---- the other BASELINE.json configs at the sizes their profiles quote (round 3) ---------------------------------------------
def _swin_task(seed=5, classes=1000):
    cfg = cls_config('swinv2_custom', classes, optimizer='AdamW', opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                     backbone_params={'img_size': 224, 'window_size': 7, 'drop_path_rate': 0.0}, inputs_shape=(3, 224, 224))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, seed)
    task.load_state_dict(sd, strict=False)
    return task.cuda()


def test_swinv2_t_b256_gradient_is_additive_over_a_batch_split_and_reproducible():
    """SwinV2-T 224 / window 7 at batch 256 (BASELINE.json configs[2] per GPU).  LayerNorm has no batch statistics, so with
    stochastic depth off the mean-loss gradient of the batch is EXACTLY the mean of the gradients of its halves in exact
    arithmetic; on the HIP path the difference is bf16 storage of the token-wise gradients + fp32 split-reduction order.
    Two runs of the same step give identical bits (fixed-order reductions, two-stream schedule)."""
    task = _swin_task().train()
    x, y = _batch(256, seed=2)

    def grads(xs, ys):
        for p in task.parameters():
            p.grad = None
        out = task.forward_with_gt({'image': xs, 'target': ys})
        task.losses(**out)[0].backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}
    full, again = grads(x, y), grads(x, y)
    assert set(full) == set(again) and all(torch.equal(full[n], again[n]) for n in full)       # bit-reproducible
    a, b = grads(x[:128], y[:128]), grads(x[128:], y[128:])
    errs = {}
    for n in full:
        ref = 0.5 * (a[n] + b[n])
        if float(ref.norm()) > 0:
            errs[n] = float((full[n] - ref).norm() / ref.norm())
    worst = max(errs, key=errs.get)
    med = sorted(errs.values())[len(errs) // 2]
    print(f'[swinv2-t B=256 batch-split additivity] {len(errs)} tensors, median {med:.2e}, worst {worst} {errs[worst]:.2e}')
    from helpers import record_distance
    record_distance('fullsize/swinv2-t B=256 batch-split additivity', 'all parameter gradients', median=med,
                    worst=errs[worst], worst_tensor=worst, tensors=len(errs))
    # bf16 activations / activation gradients: each half-batch pass rounds its own token gradients (2^-9 relative per
    # element, independent between passes); parameter gradients average that noise over >= 12544 rows
    assert med < 5e-3 and errs[worst] < 3e-2, (med, worst, errs[worst])


def test_swinv2_t_b256_branch_stream_schedule_gives_the_single_stream_bits():
    """SwinV2-T's parameter-only prologue of every attention unit (position-bias chain + qkv bias) runs on a branch stream,
    two blocks ahead (swin.py: WindowAttention.prepare), its backward on that stream too.  Same kernels, same operands:
    logits, loss and every parameter gradient must equal the single-stream schedule bit for bit — in grad mode and in
    no-grad mode, where no unit keeps the branch's output buffers alive (round 3: a qkv bias vector from the branch stream's
    pool was recycled under the main stream's queued GEMM until await_ready told the allocator about the reader)."""
    from torchok_amd.engine import core as EC
    task = _swin_task().train()
    x, y = _batch(256, seed=2)

    def step():
        for p in task.parameters():
            p.grad = None
        out = task.forward_with_gt({'image': x, 'target': y})
        loss = task.losses(**out)[0]
        loss.backward()
        torch.cuda.synchronize()
        return out['prediction'].detach().clone(), {n: p.grad.detach().clone() for n, p in task.named_parameters()
                                                    if p.grad is not None}

    def feats():
        with torch.no_grad():
            f = task.backbone.forward_features(x)
        torch.cuda.synchronize()
        return [t.clone() for t in f[1:]]
    saved = EC.BRANCH_STREAMS
    try:
        EC.BRANCH_STREAMS = False
        ref_pred, ref_grads = step()
        ref_feats = feats()
        EC.BRANCH_STREAMS = True
        for _ in range(3):
            pred, grads = step()
            assert torch.equal(pred, ref_pred)
            assert set(grads) == set(ref_grads)
            bad = [n for n in grads if not torch.equal(grads[n], ref_grads[n])]
            assert not bad, bad[:5]
            assert all(torch.equal(a, b) for a, b in zip(feats(), ref_feats))
    finally:
        EC.BRANCH_STREAMS = saved


def test_davit_t_b256_gradients_are_bit_reproducible():
    """DaViT-T 224 / window 7 at batch 256 (the size profiles/r03_davit_* quote): window attention in the plain mode, channel
    attention (Gram / apply kernels), fused Mlp, strided patch embeds — three passes of the same step give identical bits
    in the logits and in every parameter gradient."""
    cfg = cls_config('davit_t', 1000, optimizer='AdamW', opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                     backbone_params={'img_size': 224, 'window_size': 7, 'drop_path_rate': 0.0}, inputs_shape=(3, 224, 224))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 6)
    task.load_state_dict(sd, strict=False)
    task = task.cuda().train()
    x, y = _batch(256, seed=3)
    runs = []
    for _ in range(3):
        for p in task.parameters():
            p.grad = None
        out = task.forward_with_gt({'image': x, 'target': y})
        task.losses(**out)[0].backward()
        torch.cuda.synchronize()
        runs.append((out['prediction'].detach().clone(),
                     {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}))
    assert torch.isfinite(runs[0][0].float()).all()
    for pred, grads in runs[1:]:
        assert torch.equal(pred, runs[0][0])
        bad = [n for n in grads if not torch.equal(grads[n], runs[0][1][n])]
        assert not bad, bad[:5]


def test_hrnet_w48_b24_training_steps_are_bit_reproducible():
    """HRNet-W48 + neck + head at 512x1024, batch 24 (the size profiles/r0N_hrnet_* quote): two optimizer steps from the same
    state twice -> identical parameters and BatchNorm buffers (branch streams, side stream, fixed-order reductions)."""
    import bench
    res = []
    for _ in range(2):
        task = bench.build_seg_task('hrnet_w48', 19, 512, 1024)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 9)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator(device='cuda').manual_seed(4)
        x = torch.randn(24, 3, 512, 1024, generator=g, device='cuda').to(BF16)
        y = torch.randint(0, 19, (24, 512, 1024), generator=g, device='cuda')
        for it in range(2):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad(set_to_none=True)
            out['loss'].backward()
            opt.step()
        torch.cuda.synchronize()
        assert torch.isfinite(out['loss'])
        res.append((torch.cat([p.detach().reshape(-1) for p in task.parameters()]),
                    torch.cat([b.detach().double().reshape(-1) for n_, b in task.named_buffers()
                               if not n_.startswith('input_tensors')])))      # (the example input is torch.rand per construction)
        del task, opt, x, y, out
        torch.cuda.empty_cache()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize('kind', ['arcface', 'contrastive'])
def test_c5_step_at_the_real_per_rank_shape(kind):
    """BASELINE.json configs[4] per GPU: 128 x 3 x 224 x 224, ResNet-50, 11318 classes — ArcFace head + CE
    (representation_arcface_sop.yaml) and LinearHead(normalize) + ContrastiveLoss under PairwiseLearnTask (pairwise_sop.yaml).
    Properties: finite loss of the right size, every parameter gets a gradient, the relevance matrix / margin column are exact,
    and the step is bit-reproducible."""
    import math
    import bench
    res = []
    for _ in range(2):
        task = bench.build_c5_task(kind)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 21)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator(device='cuda').manual_seed(6)
        x = torch.randn(128, 3, 224, 224, generator=g, device='cuda').to(BF16)
        y = torch.randint(0, 40 if kind == 'contrastive' else 11318, (128,), generator=g, device='cuda')
        fw = task.forward_with_gt({'image': x, 'target': y})
        if kind == 'contrastive':
            R = fw['R']
            assert R.shape == (128, 128) and torch.equal(R, (y[:, None] == y[None, :]).float())        # exact
        else:
            pred = fw['prediction'].float()
            assert pred.shape == (128, 11318)
            # off the target column the logits are scale * cosine: bounded by the scale; on it the margin lowers them
            assert float(pred.abs().max()) <= task.head.scale * 1.01
        out = task.training_step({'image': x, 'target': y}, 0)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()
        assert all(p.grad is not None for p in task.parameters())
        opt.step()
        torch.cuda.synchronize()
        loss = float(out['loss'])
        assert math.isfinite(loss) and (loss > 0.5 * math.log(11318) if kind == 'arcface' else loss >= 0.0)
        res.append(torch.cat([p.detach().reshape(-1) for p in task.parameters()]))
        del task, opt, x, y, out, fw
        torch.cuda.empty_cache()
    assert torch.equal(res[0], res[1])
