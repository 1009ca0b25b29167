import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cti_amd
DEV="cuda"
torch.manual_seed(0)
B=256
v=torch.randn(B,36,2048).abs().to(DEV); q=torch.tanh(torch.randn(B,12,1024)).to(DEV).requires_grad_(True); a=torch.tanh(torch.randn(B,3,1024)).to(DEV).requires_grad_(True)
tri=cti_amd.TriAttention(2048,1024,1024,512,1,32,2,1).to(DEV).train()
tnet=cti_amd.TCNet(2048,1024,1024,512,1,32,1,k=2).to(DEV).train()
for _ in range(3):
    for prm in list(tri.parameters())+list(tnet.parameters()): prm.grad=None
    p,_=tri(v,q,a)
    o=tnet.forward_with_weights(v,q,a,p[...,0])
    o.sum().backward()
torch.cuda.synchronize()
