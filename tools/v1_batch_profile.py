import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_vlp16_900(0.0)
rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(model)
rcc.find(T.identity()); mv = rcc.modelView()
rcc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
rng = np.random.RandomState(1)
poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-0.5, 0.5, 3)), (0, 0, rng.uniform(-0.1, 0.1))) for _ in range(1000)], dtype=T.TRANSFORM)
for _ in range(3): rcc.correct_batch(poses)
t = time.perf_counter()
for _ in range(20): rcc.correct_batch(poses)
print("ms per batch", (time.perf_counter() - t) / 20 * 1e3)
