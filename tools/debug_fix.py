import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import kplanes_oracle as KO
from soccernerfs_amd.trainer import KPlanesTrainer, KPlanesTrainConfig
DEV="cuda:0"
E = dict(base_res=(16, 16, 16, 4), multiscale=(1, 2), feat_dim=32, prop_res=((24, 24, 24, 4), (32, 32, 32, 4)), prop_feat=8, sigma_hidden=128, color_hidden=64, aabb_scale=1.5, seed=5)
P = KO.make_kplanes_params(**E)
for sc in P["field_grids"]:
    sc[0].zero_()
R = 40
gen = torch.Generator().manual_seed(3)
dv = lambda z: z.to(DEV).contiguous()
rays = {"origins": dv((torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2), "directions": dv(torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)), "times": dv(torch.rand(R, 1, generator=gen))}
target = dv(torch.rand(R, 3, generator=gen))
rng = {"t_rand": dv(torch.rand(R, 65, generator=gen)), "u": [dv(torch.rand(R, 33, generator=gen)), dv(torch.rand(R, 17, generator=gen))], "bg": dv(torch.rand(R, 3, generator=gen))}
cfg = KPlanesTrainConfig(aabb_scale=1.5, spacetime_resolution=E["base_res"], multiscale_res=E["multiscale"], feature_dim=32, proposal_resolutions=E["prop_res"], proposal_feature_dim=8,
                         num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16, warm_up_end=2, fix_capacity=4)
small = KPlanesTrainer(cfg, R, DEV)
small.load_oracle_params(P)
print("cap", small._ss.fix_capacity, small._ss.N * small._ss.ps.out_dim, small.quotient_scatter, small.quotient_epilogue)
for i in range(18):
    small.forward(rays, rng, 1.0, training=True)
    small.backward(target, rng, proposal_grads=True, include_reg=True)
    torch.cuda.synchronize()
    print(i, "qg", small._qg_step, "counts", small._ss.fix_counts.tolist(), "peak", small._ss.fix_peak.item(), "host", int(small._fix_peak_host[0]), "step", small.step, flush=True)
    try:
        small.optimizer_step()
    except RuntimeError as e:
        print("RAISED", str(e)[:80]); break
    torch.cuda.synchronize()
