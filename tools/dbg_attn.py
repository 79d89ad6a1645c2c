import sys, torch
sys.path.insert(0,'.')
from eda_amd import attention
from oracle import attention_ref
torch.manual_seed(0)
B,Lq,Lk=2,16,32
q=torch.randn(B,Lq,288,device='cuda'); k=torch.randn(B,Lk,288,device='cuda'); v=torch.randn(B,Lk,288,device='cuda')
lens=torch.tensor([Lk,10]); mask=(torch.arange(Lk)[None,:]>=lens[:,None]).cuda()
o_m=attention.attention_core(q,k,v,mask,8,0.0,0)
o_n=attention.attention_core(q,k,v,None,8,0.0,0)
e_m=attention_ref.attention_core(q,k,v,mask,8)
e_n=attention_ref.attention_core(q,k,v,None,8)
e_t=attention_ref.attention_core(q[1:],k[1:,:10],v[1:,:10],None,8)
print("masked vs ref-masked", (o_m-e_m).abs().amax(dim=(1,2)))
print("masked vs ref-unmasked", (o_m-e_n).abs().amax(dim=(1,2)))
print("unmasked vs ref-unmasked", (o_n-e_n).abs().amax(dim=(1,2)))
print("ref masked vs ref trunc", (e_m[1:]-e_t).abs().max())
# per-key experiment: which keys are treated dead? use V = one-hot of key index in dim 0..31
vv=torch.zeros(B,Lk,288,device='cuda')
for kk in range(Lk): vv[:,kk,kk]=1.0   # head 0 dims 0..31 -> prob of key kk (Lk<=32)
qq=torch.zeros_like(q)  # uniform probs
p=attention.attention_core(qq,k,vv,mask,8,0.0,0)[1,0,:32]
print("probs batch1 q0 head0:", p.cpu().numpy().round(3))
print("mask row1:", mask[1].int().cpu().numpy())
