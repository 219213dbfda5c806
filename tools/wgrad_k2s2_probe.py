"""GPU-box aid: fetched bytes of ONE stride-2 weight-gradient launch (vs_conv_wgrad_multi, kind K2S2) by which operand is lazy and how the tensors are aligned.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -o p -- python3 tools/wgrad_k2s2_probe.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_segmentation_amd import ops
from vae_segmentation_amd._lib import VS_CONV_K2S2, check, lib

n, s, C = 2, 48, 16
gen = torch.Generator().manual_seed(0)
dt = torch.bfloat16
pbuf = torch.randn(n * s ** 3 * C + 256, generator=gen).to(dt).cuda()
qbuf = torch.randn(n * (2 * s) ** 3 * C + 256, generator=gen).to(dt).cuda()
ops.stats_arena_begin(pbuf.device)
dw = torch.empty(C, C, 8, dtype=torch.float32, device="cuda")
st = ops._stream()
for name, poff, qoff, plazy, qlazy in (("P lazy", 0, 0, 1, 0), ("Q lazy", 0, 0, 0, 1), ("none lazy", 0, 0, 0, 0), ("Q +64 B", 0, 32, 1, 0), ("P lazy again", 0, 0, 1, 0)):
    p = pbuf[poff:poff + n * s ** 3 * C].view(n, s, s, s, C)
    q = qbuf[qoff:qoff + n * (2 * s) ** 3 * C].view(n, 2 * s, 2 * s, 2 * s, C)
    ps = ops.instnorm_stats(p) if plazy else None
    qs = ops.instnorm_stats(q) if qlazy else None
    d = ops.WgradDesc(p.data_ptr(), ps.data_ptr() if plazy else None, q.data_ptr(), qs.data_ptr() if qlazy else None, dw.data_ptr(), None, None, 0, 0, 0,
                      n, s, s, s, C, C, C, C, VS_CONV_K2S2, 0)
    arr = (ops.WgradDesc * 1)(d)
    nb = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(arr), 1, ops.vs_dtype(p))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        check(lib.vs_conv_wgrad_multi(ctypes.addressof(arr), 1, ws.data_ptr(), nb, ops.vs_dtype(p), 1e-5, st), "multi")
    torch.cuda.synchronize()
    print(name, "p %x q %x" % (p.data_ptr(), q.data_ptr()))
