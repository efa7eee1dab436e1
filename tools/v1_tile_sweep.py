import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import rmcl_amd as ra
from rmcl_amd import synthetic as syn, types as T
ctx = ra.Context(0)
v, f = syn.uv_sphere(100000)
hm = ra.import_hip_map(ctx, v, f)
model = syn.model_vlp16_900(0.0)
rng = np.random.RandomState(1)
poses = np.array([T.transform_from_rpy(tuple(rng.uniform(-1.0, 1.0, 3) * (1, 1, 0.3)), (0, 0, rng.uniform(-3, 3))) for _ in range(1000)], dtype=T.TRANSFORM)
for tile_bits in (0, 3, 4, 5, 6, 7):
    rcc = ra.RCCHipSpherical(hm); rcc.setTsb(T.identity()); rcc.setModel(model)
    rcc.find(T.identity()); mv = rcc.modelView()
    rcc.set_dataset(mv["points"].reshape(-1, 3), mv["hits"].reshape(-1))
    rcc.set_variant(15 | (tile_bits << 4))
    for _ in range(3): rcc.correct_batch(poses)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(5): rcc.correct_batch(poses)
        ts.append((time.perf_counter() - t) / 5 * 1e3)
    print("tile_bits %d (width %s): %.4f ms per batch" % (tile_bits, "auto" if tile_bits == 0 else str(1 << (tile_bits - 1)), sorted(ts)[2]), flush=True)
    rcc.close()
