import sys; sys.path.insert(0, "/root/repo")
import torch, ctypes as C
from ultrafnd_git_amd.trainer import ForensicTrainer, TrainConfig, synthetic_cache
from ultrafnd_git_amd import _lib as L
B=4
cfg = TrainConfig(data_root="", ocr_phrase_pkl=None, out_dir="/tmp/dbg", batch_size=B, device="cuda", use_graph=False)
tr = ForensicTrainer(cfg, cache=synthetic_cache(16, seed=1))
tr.fusion.train(); tr.clf.train()
it = iter(tr.train_loader); batch = next(it)
b = tr.head.bufs(B, True)
tr._load_batch(b, batch, "train")
tr.head.enqueue_forward(b, B, True, True)
torch.cuda.synchronize()
H=512
off = B*16*H + B*9*H
ev = b["fws"].view(-1)[off:off+B*4].clone()
print("evid after fwd", ev.tolist())
tr.arena.grad.fill_(7.0)
tr.head.enqueue_backward(b, B, 0)
torch.cuda.synchronize()
print("evid after bwd", b["fws"].view(-1)[off:off+B*4].tolist())
for k in ("attn_tv","attn_ta","attn_vu"):
    print(k, tr.arena.grad_view(f"fusion.{k}.evidence_proj.0.weight").norm().item(), tr.arena.grad_view(f"fusion.{k}.evidence_proj.2.weight").norm().item())

for k in ("attn_tv","attn_ta","attn_vu"):
    g0=tr.arena.grad_view(f"fusion.{k}.evidence_proj.0.weight"); print(k, "0.weight first rows", g0.flatten()[:9].tolist(), "count==7:", int((g0==7.0).sum()), "of", g0.numel())
    g2=tr.arena.grad_view(f"fusion.{k}.evidence_proj.2.weight"); print(k, "2.weight", g2.flatten()[:4].tolist(), int((g2==7.0).sum()))

def al64(n): return (n+63)//64*64
sizes=[B*16*H,B*9*H,B*4,B*4,B*4,16*B*2*H,B*2*H,B*2*H,B*H,B*H,B*2*H,4*B*16*H,4*B*H,B*H,B*9*H]
o=sum(al64(x) for x in sizes)
print("dout", b["fws"].view(-1)[o:o+B*4].tolist())
print("gate", b["fws"].view(-1)[al64(B*16*H)+al64(B*9*H)+64: al64(B*16*H)+al64(B*9*H)+64+16].tolist())
