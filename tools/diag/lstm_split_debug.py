import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl8_amd import hip
DEV="cuda:0"
b,l,d = (int(a) for a in sys.argv[1:4]) if len(sys.argv)>3 else (127,4,1)
g = torch.Generator(device=DEV).manual_seed(100*b+10*l+d)
lstm = torch.nn.LSTM(d, 256, batch_first=True).to(DEV)
with torch.no_grad():
    for p in lstm.parameters(): p.copy_(torch.randn(p.shape, device=DEV, generator=g)*0.2)
x = torch.randn(b,l,d,device=DEV,generator=g)*2; h0=torch.randn(b,256,device=DEV,generator=g)*0.5; c0=torch.randn(b,256,device=DEV,generator=g)
params=(lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0)
ref = hip.lstm_forward(x,h0,c0,hip.lstm_pack(*params),save=True)
packed, wb = hip.lstm_pack_split(*params)
got = hip.lstm_forward_split(x,h0,c0,packed,wb,save=True)
torch.cuda.synchronize()
for name,a,r in zip(("hs","hn","cn","gates","cs"),got,ref):
    dd=(a-r).abs(); i=dd.argmax(); idx=[int(v) for v in torch.unravel_index(i, dd.shape)]
    print(name, "max abs", float(dd.max()), "at", idx, "got", float(a.flatten()[i]), "ref", float(r.flatten()[i]), "count>1e-5", int((dd>1e-5).sum()))
