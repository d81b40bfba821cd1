"""OCRSegmentationHead (second half) against oracle/ocr_ref.py and tests/golden/ocr_head_step.npz
(forward + backward of the reference's OWN heads/segmentation/ocr.py in training mode, tests/golden/gen_golden.py).
Each test runs on the host stand-in and, marked gpu, through libtok_gfx950.so."""
import os

import numpy as np
import pytest
import torch

import oracle.ocr_ref as O
import torchok_amd as T
from helpers import deterministic_state, rel_err

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ocr_head_step.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _head():
    head = T.HEADS.get('OCRSegmentationHead')(in_channels=int(GOLD['in_channels']), num_classes=int(GOLD['num_classes']))
    ref = O.OCRSegmentationHead(int(GOLD['in_channels']), int(GOLD['num_classes']))
    assert {k: tuple(v.shape) for k, v in head.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    head.load_state_dict(deterministic_state(ref.state_dict(), int(GOLD['seed'])))
    return head


def test_training_forward_backward_vs_reference_golden(dev, monkeypatch):
    from torchok_amd.models.heads import ocr_head as OH
    head = _head().to(dev).train()
    monkeypatch.setattr(OH.SpatialOCR, 'draw_dropout',
                        lambda self, batch, channels, device: torch.from_numpy(GOLD['drop_scale']).to(device))
    hw = tuple(int(v) for v in GOLD['image_hw'])
    feats = torch.from_numpy(GOLD['feats']).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) \
        .requires_grad_(True)                      # what a torchok_amd neck hands over
    image = torch.zeros(feats.shape[0], 3, *hw, device=dev)
    out, aux = head([image, feats])
    assert tuple(out.shape) == GOLD['out'].shape and tuple(aux.shape) == GOLD['out_aux'].shape
    assert rel_err(aux.float(), torch.from_numpy(GOLD['out_aux']).float()) < 2e-2
    assert rel_err(out.float(), torch.from_numpy(GOLD['out']).float()) < 4e-2
    g = torch.Generator().manual_seed(int(GOLD['weight_seed']))          # the generator's draws: feats, w_out, w_aux
    assert torch.equal(torch.randn(GOLD['feats'].shape, generator=g).half().float(), torch.from_numpy(GOLD['feats']))
    wo, wa = (torch.randn(GOLD['out'].shape, generator=g).to(dev) for _ in range(2))
    ((out.float() * wo).sum() + 0.4 * (aux.float() * wa).sum()).backward()
    params = dict(head.named_parameters())
    assert all(p.grad is not None for p in params.values())
    names = [str(n) for n in GOLD['param_names']]
    gn = np.array([float(params[n].grad.detach().double().norm()) for n in names])
    assert np.median(np.abs(gn / GOLD['grad_norm'] - 1)) < 0.05, np.abs(gn / GOLD['grad_norm'] - 1)
    # bf16 through BatchNorm + ReLU units over 1536 pixels moves these gradients by 20-30 % for ANY bf16 implementation
    # (mask flips in the 8-channel last_reduction): torch's own bf16-autocast run of the oracle is the yardstick
    ref = O.OCRSegmentationHead(int(GOLD['in_channels']), int(GOLD['num_classes'])).train()
    ref.load_state_dict(deterministic_state(ref.state_dict(), int(GOLD['seed'])))
    monkeypatch.setattr(torch.nn.functional, 'dropout2d',
                        lambda x, p=0.5, training=True, inplace=False: x * torch.from_numpy(GOLD['drop_scale'])[:, :, None, None])
    xa = torch.from_numpy(GOLD['feats']).requires_grad_(True)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        oa, aa = ref([image.cpu(), xa])
    ((oa.float() * wo.cpu()).sum() + 0.4 * (aa.float() * wa.cpu()).sum()).backward()
    yard = dict(ref.named_parameters())
    small = [n for n in names if 'grad__' + n in GOLD.files]
    errs = {n: rel_err(params[n].grad.float(), torch.from_numpy(GOLD['grad__' + n])) for n in small}
    ycap = {n: rel_err(yard[n].grad.float(), torch.from_numpy(GOLD['grad__' + n])) for n in small}
    assert np.median(list(errs.values())) < 1.5 * np.median(list(ycap.values())) + 1e-2
    bad = [n for n in small if errs[n] > 1.5 * ycap[n] + 0.05]
    assert len(bad) <= 2, [(n, errs[n], ycap[n]) for n in bad]
    assert rel_err(feats.grad.float(), torch.from_numpy(GOLD['d_feats'])) < 1.5 * rel_err(xa.grad, torch.from_numpy(GOLD['d_feats'])) + 1e-2


def test_eval_returns_one_tensor(dev):
    """The golden eval output was taken after ONE training forward (running statistics updated once)."""
    head = _head().to(dev).train()
    hw = tuple(int(v) for v in GOLD['image_hw'])
    feats = torch.from_numpy(GOLD['feats']).to(dev)
    image = torch.zeros(feats.shape[0], 3, *hw, device=dev)
    torch.manual_seed(0)                                   # the channel-dropout draw of the training pass
    with torch.no_grad():
        assert len(head([image, feats])) == 2
        out = head.eval()([image, feats])
    assert isinstance(out, torch.Tensor) and rel_err(out.float(), torch.from_numpy(GOLD['eval_out']).float()) < 8e-2


def test_limits(dev):
    with pytest.raises(NotImplementedError):
        T.HEADS.get('OCRSegmentationHead')(in_channels=16, num_classes=40, ocr_mid_channels=512).to(dev).train()(
            [torch.zeros(1, 3, 16, 16, device=dev), torch.randn(2, 16, 4, 4, device=dev)])     # 40 x 512 > 10240
