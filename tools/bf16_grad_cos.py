import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "multimodal-baby_amd"))
import torch, torch.nn.functional as F
from multimodal.resnext import ResNet
dev = torch.device("cuda:0")
for B, S in [(8, 64), (16, 128), (32, 224)]:
    torch.manual_seed(0)
    model = ResNet().to(dev).train()
    x = torch.randn(B, 3, S, S, device=dev)
    dp = torch.randn(B, 2048, device=dev)
    grads = {}
    for cdt in (torch.float32, torch.bfloat16):
        model.compute_dtype = cdt
        model.zero_grad(set_to_none=True)
        pooled, _ = model.trunk(x)
        pooled.backward(dp)
        grads[cdt] = ({k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}, pooled.detach().clone())
    cs = []
    for k in grads[torch.float32][0]:
        a, b = grads[torch.float32][0][k].flatten().double(), grads[torch.bfloat16][0][k].flatten().double()
        cs.append((float(F.cosine_similarity(a, b, dim=0)), k))
    cs.sort()
    pc = float(F.cosine_similarity(grads[torch.float32][1].flatten().double(), grads[torch.bfloat16][1].flatten().double(), dim=0))
    print(B, S, "pooled cos", round(pc, 4), "grad cos min", cs[:3], "median", cs[len(cs)//2], "max", cs[-1])
