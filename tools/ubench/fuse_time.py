import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from torchok_amd import _C
lib = _C.load_library()
st = torch.cuda.current_stream().cuda_stream
BF = torch.bfloat16
for (n, h, w, c, shifts) in [(24, 128, 256, 48, (0, 1, 2, 3)), (24, 64, 128, 96, (0, 0, 1, 2)), (24, 32, 64, 192, (0, 0, 0, 1)), (24, 16, 32, 384, (0, 0, 0, 0))]:
    terms = [torch.randn(n, h >> s, w >> s, c, device='cuda').to(BF) for s in shifts]
    sc = [torch.rand(c, device='cuda') + 0.5 for _ in shifts]
    sf = [torch.randn(c, device='cuda') for _ in shifts]
    out = torch.empty(n, h, w, c, device='cuda', dtype=BF)
    mask = torch.empty(n * h * w, c // 8, device='cuda', dtype=torch.uint8)
    tmp = [torch.empty_like(t) for t in terms]
    def run(mode):
        if mode == 'affine':
            a = []
            for i in range(4):
                a += [terms[i].data_ptr(), shifts[i]] + ([sc[i].data_ptr(), sf[i].data_ptr()] if i > 0 else [None, None])
            lib.tok_fuse_sum_affine_relu_fwd(*a, n, h, w, c, 1, out.data_ptr(), mask.data_ptr(), st)
        else:
            for i in range(1, 4):      # the apply passes of the three path terms, then the plain sum
                m = terms[i].numel() // c
                lib.tok_bn_act_fwd(terms[i].data_ptr(), sc[i].data_ptr(), sf[i].data_ptr(), None, 0, tmp[i].data_ptr(), None, m, c, st)
            a = [terms[0].data_ptr(), shifts[0]]
            for i in range(1, 4):
                a += [tmp[i].data_ptr(), shifts[i]]
            lib.tok_fuse_sum_relu_fwd(*a, n, h, w, c, 1, out.data_ptr(), mask.data_ptr(), st)
    for mode in ('affine', 'apply+sum'):
        for _ in range(3): run(mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(mode)
        e1.record(); torch.cuda.synchronize()
        print((n, h, w, c, shifts), mode, f'{e0.elapsed_time(e1) * 50:.1f} us')
