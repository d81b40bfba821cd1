import os, sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import torchok_amd as T
from helpers import cls_config, deterministic_state
for name in ['resnet18_cls_step', 'resnet50_cls_step']:
    g = np.load(os.path.join('tests/golden', name + '.npz'))
    backbone, classes, seed = str(g['backbone']), int(g['num_classes']), int(g['seed'])
    cfg = cls_config(backbone, classes)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, seed)
    task.load_state_dict(sd, strict=False)
    task.cuda().train()
    x, y = torch.from_numpy(g['x'].astype(np.float32)).cuda(), torch.from_numpy(g['y']).cuda()
    feats = task.backbone.forward_features(x)
    fe = [abs(float((f.detach().double() ** 2).sum().item()) / float(ss) - 1) for f, ss in zip(feats[1:], g['feat_sumsq'][1:])]
    out = task.training_step({'image': x, 'target': y}, 0)
    fw = task.forward_with_gt({'image': x, 'target': y})
    pred = fw['prediction'].detach().float().cpu().numpy()
    pe = np.linalg.norm(pred - g['prediction']) / np.linalg.norm(g['prediction'])
    le = abs(float(out['loss'].detach().item()) - float(g['loss'])) / abs(float(g['loss']))
    out['loss'].backward()
    gn = np.array([float(p.grad.detach().double().norm().item()) for _, p in task.named_parameters()])
    r = np.abs(gn / g['grad_norm'] - 1)
    fcb = task.head.fc.bias.grad.detach().float().cpu().numpy()
    fe2 = np.abs(fcb - g['grad__head.fc.bias']).max() / np.abs(g['grad__head.fc.bias']).max()
    print(name, 'x', tuple(x.shape), 'feat', ['%.4f' % v for v in fe], 'pred %.4f loss %.5f gradnorm median %.4f p90 %.4f max %.4f fcb %.4f' % (pe, le, np.median(r), np.percentile(r, 90), r.max(), fe2))
