"""dec_logits_kernel against dec_linear_kernel<ARGMAX>: the (max, index) workspace of one call, saved to a file per setting of
PM_DEC_LOGITS_PERSIST and compared by the second run.   python tools/logits_check.py OUT.pt [REF.pt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib
L = lib()
torch.manual_seed(0)
res = {}
for (M, K, V) in [(32, 512, 51865), (2, 512, 51865), (17, 384, 51865), (32, 512, 1000), (32, 256, 50257)]:
    x = torch.randn(M, K, device="cuda") * 3
    g = torch.rand(K, device="cuda") + 0.5
    b = torch.randn(K, device="cuda") * 0.1
    E = (torch.randn(V, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    tile = L.pm_dec_argmax_tile(K)
    nt = (V + tile - 1) // tile
    wv = torch.full((M, nt), float("nan"), device="cuda")
    wi = torch.full((M, nt), -7, dtype=torch.int32, device="cuda")
    rc = L.pm_dec_linear(x.data_ptr(), K, g.data_ptr(), b.data_ptr(), 1e-5, E.data_ptr(), K, None, None, 0, None, 0, M, V, K, 0, 2,
                         None, None, 0, 0, 0, None, wv.data_ptr(), wi.data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    res[(M, K, V)] = (wv.cpu(), wi.cpu())
    print((M, K, V), "nan:", int(torch.isnan(wv).sum()), "unset idx:", int((wi == -7).sum()), "idx range", int(wi.min()), int(wi.max()))
torch.save(res, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k in res:
        same_v = torch.equal(res[k][0], ref[k][0]); same_i = torch.equal(res[k][1], ref[k][1])
        print(k, "values equal:", same_v, "indices equal:", same_i)
        if not same_v:
            d = (res[k][0] != ref[k][0]).nonzero()
            print("  first differences at", d[:5].tolist(), res[k][0][tuple(d[0])].item(), ref[k][0][tuple(d[0])].item())
