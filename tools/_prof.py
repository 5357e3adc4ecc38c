import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cra5_amd import synth
from cra5_amd.zoo import vaeformer_pretrained
dev = torch.device("cuda:0")
net = vaeformer_pretrained(quality=268, pretrained=False); synth.load_synthetic(net, seed=7); net = net.to(dev)
net.precision = sys.argv[1] if len(sys.argv) > 1 else "f16"
net.gpu_exclusive = False
x = synth.synth_frame(268, 1000).unsqueeze(0).to(dev)
for _ in range(3):
    out = net.compress(x); rec = net.decompress(out["strings"], out["z_shape"])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    out = net.compress(x); rec = net.decompress(out["strings"], out["z_shape"])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28); print(s.getvalue()[:6000])
