"""Edge cases of the built rows: ragged spatial sizes, tiny batches, single-channel inputs, one-class heads, wrong image
sizes, frozen parameters — each against the oracle (values) or the reference's documented behaviour (errors).  Host
stand-in and, marked gpu, libtok_gfx950.so."""
import pytest
import torch

import oracle.torchok_ref as R
import torchok_amd as T
from helpers import cls_config, copy_state, deterministic_state, rel_err


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _pair(backbone, classes, seed=3, **bk):
    cfg = cls_config(backbone, classes, backbone_params=bk or None)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    ref = R.ClassificationModel(backbone, classes, **bk)
    ref.load_state_dict(deterministic_state(ref.state_dict(), seed))
    copy_state(ref, task)
    return task, ref


@pytest.mark.parametrize('shape', [(3, 3, 33, 47), (1, 3, 64, 64), (2, 3, 17, 17), (5, 3, 224, 96)])
def test_ragged_sizes_and_tiny_batches_eval(dev, shape):
    """Odd heights / widths (every conv, the 3x3/s2 max-pool and the stride-2 dgrad see ragged maps) and batch 1, eval mode
    (running statistics) so a single image is well defined."""
    task, ref = _pair('resnet18', 7)
    task.to(dev).eval()
    ref.eval()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = task(x.to(dev))
        yr = ref.forward_with_gt({'image': x, 'target': None})['prediction']
    assert y.shape == yr.shape
    assert rel_err(y.float(), yr) < 3e-2


def test_ragged_training_step_gradients(dev):
    task, ref = _pair('resnet18', 7)
    task.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(2)
    x, y = torch.randn(6, 3, 45, 59, generator=g), torch.randint(0, 7, (6,), generator=g)
    out = task.training_step({'image': x.to(dev), 'target': y.to(dev)}, 0)
    out['loss'].backward()
    loss_ref, _ = R.training_step(ref, {'image': x, 'target': y}, None)
    assert abs(float(out['loss'].detach()) - float(loss_ref)) < 3e-2 * max(1.0, float(loss_ref))
    rp = dict(ref.named_parameters())
    assert rel_err(task.head.fc.weight.grad, rp['head.fc.weight'].grad) < 0.15
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in task.parameters())


def test_single_channel_input_and_one_class_head(dev):
    """in_channels=1 (resnet.py:488 takes `in_channels`), num_classes=1 squeezes the prediction (classification_head.py:38-39)."""
    task, ref = _pair('resnet18', 1, in_channels=1)
    task.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 1, 64, 64, generator=g)
    fw = task.forward_with_gt({'image': x.to(dev), 'target': torch.zeros(8)})
    assert fw['prediction'].shape == (8,)
    pr = ref.forward_with_gt({'image': x, 'target': None})['prediction']
    assert rel_err(fw['prediction'].float(), pr) < 5e-2


def test_frozen_backbone_only_trains_the_head(dev):
    task, _ = _pair('resnet18', 4)
    for p in task.backbone.parameters():
        p.requires_grad_(False)
    task.to(dev).train()
    out = task.training_step({'image': torch.randn(4, 3, 64, 64).to(dev), 'target': torch.randint(0, 4, (4,)).to(dev)}, 0)
    out['loss'].backward()
    assert all(p.grad is None for p in task.backbone.parameters())
    assert all(p.grad is not None for p in task.head.parameters())


def test_swin_rejects_a_wrong_image_size_and_hrnet_needs_multiples_of_32(dev):
    sw = T.BACKBONES.get('swinv2_custom')(img_size=64, window_size=4, depths=(2, 2, 2, 2)).to(dev)
    with pytest.raises(AssertionError, match="doesn't match model"):       # [timm] PatchEmbed.forward
        sw(torch.rand(1, 3, 96, 96).to(dev))
    hr = T.BACKBONES.get('hrnet_w18_small')(pretrained=False).to(dev)
    with pytest.raises(ValueError):
        hr(torch.rand(1, 3, 40, 40).to(dev))
    with pytest.raises(KeyError):
        T.BACKBONES.get('resnet51')
    with pytest.raises(RuntimeError, match='pretrained'):
        T.BACKBONES.get('resnet18')(pretrained=True)


@pytest.mark.parametrize('shape', [(6, 5), (2, 5, 4, 4), (20000, 16)])
def test_cross_entropy_out_of_range_labels_are_dropped_consistently(dev, shape):
    """A label that is neither a class index nor `ignore_index` (a 255 'void' pixel with the default ignore_index=-100):
    torch raises; the kernels drop the row from the loss, from the mean denominator AND from the gradient — the value
    equals torch's with those rows ignored (one predicate in forward / partial / backward, wave-per-row and thread-per-row
    kernels)."""
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(*shape, generator=g).to(torch.bfloat16)
    tshape = (shape[0],) + tuple(shape[2:])
    target = torch.randint(0, shape[1], tshape, generator=g)
    target.view(-1)[::3] = 255
    target.view(-1)[1] = -7
    z = logits.to(dev).detach().clone().requires_grad_(True)
    loss = T.LOSSES.get('CrossEntropyLoss')()(input=z, target=target.to(dev))
    loss.backward()
    zr = logits.float().detach().clone().requires_grad_(True)
    clean = torch.where((target < 0) | (target >= shape[1]), torch.full_like(target, -100), target)
    want = torch.nn.functional.cross_entropy(zr, clean, ignore_index=-100)
    want.backward()
    assert abs(float(loss.detach()) - float(want.detach())) < 2e-5 * max(1.0, abs(float(want.detach())))
    assert rel_err(z.grad.float().cpu(), zr.grad) < 5e-3
    dropped = ((target < 0) | (target >= shape[1]))
    gd = z.grad.float().cpu().movedim(1, -1)[dropped] if len(shape) == 4 else z.grad.float().cpu()[dropped]
    assert float(gd.abs().max()) == 0.0
