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
