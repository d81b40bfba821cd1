"""The further ResNet entrypoints of the reference (resnet.py:597-776, 881-960): other depths, the wide bottleneck
(base_width 128) and the 'd' / 't' variants (deep 3x3 stem, average-pool shortcut projections).  Same blocks and kernels as resnet18/50, so the checks are: the parameter tree equals
the oracle's (timm's BasicBlock / Bottleneck restated in oracle/timm_min.py), and a training step of the two new
shapes of layer — [2,2,2,2] bottlenecks and 2x-wide 3x3s — stays inside the bf16 yardstick of the fp32 oracle."""
import copy

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


def test_registered_names_and_parameter_trees(fake_backend):
    names = ['resnet18d', 'resnet34d', 'resnet26d', 'resnet26t', 'resnet50d', 'resnet50t', 'resnet101d', 'resnet152d',
             'resnet200d', 'resnet26', 'resnet200', 'tv_resnet34', 'tv_resnet50', 'tv_resnet101', 'tv_resnet152', 'wide_resnet50_2',
             'wide_resnet101_2', 'ssl_resnet18', 'ssl_resnet50', 'swsl_resnet18', 'swsl_resnet50']
    assert all(n in T.BACKBONES.entrypoints for n in names)
    for n in ('resnet18d', 'resnet26t', 'resnet50d', 'resnet26', 'wide_resnet50_2', 'tv_resnet34', 'tv_resnet50', 'ssl_resnet18', 'swsl_resnet50'):
        mine = T.BACKBONES.get(n)(pretrained=False, in_channels=3)
        ref = R.BACKBONES[n]()
        assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == \
            {k: tuple(v.shape) for k, v in ref.state_dict().items()}, n
        assert mine.out_channels == ref.out_channels
    assert T.BACKBONES.get('wide_resnet50_2')(pretrained=False).layer1[0].conv2.weight.shape == (128, 128, 3, 3)
    depth = lambda m: sum(len(getattr(m, f'layer{i}')) for i in range(1, 5))      # noqa: E731
    assert depth(T.BACKBONES.get('resnet200')(pretrained=False)) == 66
    assert depth(T.BACKBONES.get('wide_resnet101_2')(pretrained=False)) == 33
    with pytest.raises(RuntimeError):
        T.BACKBONES.get('ssl_resnet18')()          # the reference default asks for a download (pretrained=True)


# resnet18d / resnet26t at 72 px: 18 -> 9 -> 5 -> 3 feature maps, so the ceil-mode average pool meets odd sizes
@pytest.mark.parametrize('backbone,size', [('resnet26', 64), ('wide_resnet50_2', 64), ('resnet18d', 72), ('resnet26t', 72)])
def test_training_step_vs_oracle(dev, backbone, size):
    torch.manual_seed(0)
    cfg = cls_config(backbone, 10, backbone_params={'zero_init_last': False})
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    ref = R.ClassificationModel(backbone, 10, zero_init_last=False)
    ref.load_state_dict(deterministic_state(ref.state_dict(), 23))
    copy_state(ref, task)
    task.to(dev).train()
    ref.train()
    x, y = torch.randn(16, 3, size, size), torch.randint(0, 10, (16,))
    ref2 = copy.deepcopy(ref)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o = ref2.forward_with_gt({'image': x, 'target': y})
    ac_loss = torch.nn.functional.cross_entropy(o['prediction'].float(), y)
    ac_loss.backward()
    yard = {n: p.grad for n, p in ref2.named_parameters()}
    out = task.training_step({'image': x.to(dev), 'target': y.to(dev)}, 0)
    out['loss'].backward()
    ref_loss, _ = R.training_step(ref, {'image': x, 'target': y}, None)
    assert abs(float(out['loss'].detach()) - float(ref_loss)) < max(2e-2, 1.5 * abs(float(ac_loss.detach()) - float(ref_loss)) + 1e-2)
    rp = dict(ref.named_parameters())
    for n, p in task.named_parameters():
        assert p.grad is not None, n
        assert rel_err(p.grad, rp[n].grad) < 1.5 * rel_err(yard[n], rp[n].grad) + 1e-2, n
