import sys, types
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from etch_amd import constants as K
from etch_amd.models.models_pointcloud import GT_network_equiv
from etch_amd.utils.weights import load_seeded, seeded_state_dict
from oracle import stage1 as S1
args = types.SimpleNamespace(output_folder="/tmp/tiny", EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"), markerset=K.default_markerset(), scale_magnitude=10)
model = load_seeded(GT_network_equiv(option=args), 1).cuda().eval()
sd = {k: v.cpu() for k, v in seeded_state_dict(model, 1).items()}
for B, N in ((1, 130), (2, 257), (1, 256), (1, 300), (3, 511)):
    pts = torch.from_numpy(np.stack([(np.random.default_rng(7 + b).standard_normal((N, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32) for b in range(B)]))
    try:
        with torch.no_grad():
            res, _ = model(pts.cuda(), ["confidence", "direction", "magnitude"], "standard_vector")
        torch.cuda.synchronize()
    except Exception as e:
        print(B, N, "GPU path raised", type(e).__name__, str(e)[:120]); continue
    try:
        ref = S1.forward(sd, pts, S1.build_layer_table(), return_aux=True)
    except Exception as e:
        print(B, N, "oracle raised", type(e).__name__, str(e)[:120]); continue
    errs = {k: float((res[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()) for k in ("part_labels", "confidences", "magnitude")}
    errs["anc_w"] = float((model.last_anc_w.cpu() - ref["anc_w"]).abs().max() / ref["anc_w"].abs().max())
    print(B, N, {k: f"{v:.2e}" for k, v in errs.items()})
