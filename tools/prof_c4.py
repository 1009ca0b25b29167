import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cti_amd
DEV="cuda"
torch.manual_seed(0)
B=256
v=torch.randn(B,36,2048).abs().to(DEV); q=torch.tanh(torch.randn(B,12,1024)).to(DEV); a=torch.tanh(torch.randn(B,3,1024)).to(DEV)
bi=cti_amd.BiAttention(2048,1024,1024,8).to(DEV).eval()
tri=cti_amd.TriAttention(2048,1024,1024,512,1,32,2,1).to(DEV).eval()
tnet=cti_amd.TCNet(2048,1024,1024,512,1,32,1,k=2).to(DEV).eval()
with torch.no_grad():
    for _ in range(5):
        pb,_=bi.forward_all(v,q)
        p,_=tri(v,q,a)
        o=tnet.forward_with_weights(v,q,a,p[...,0])
torch.cuda.synchronize()
