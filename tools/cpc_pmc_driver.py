"""100 cold closest-point finds on the C2 dataset (sphere-100k) for a PMC pass: `grid` seeds every query from the map's near grid
(the round-4 default), `bare` does not (round 3's cold query).  profiles/r04_pmc_cpc_cold.txt (tools/pmc_sets.sh)."""
import sys
sys.path.insert(0, '/root/repo')
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
mode = sys.argv[1]
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
truth = syn.pose_c2_truth()
est = T.mult(truth, syn.pose_c2_perturbation())
rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(syn.model_c2()); rcc.find(truth); mv = rcc.modelView()
cpc = ra.CPCHip(hm); cpc.setTsb(T.identity()); cpc.params.max_dist = 1.0
cpc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
cpc.set_tracking(False); cpc.set_grid(mode == "grid")
for _ in range(100): cpc.find(est)
cpc.close()
