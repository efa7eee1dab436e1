"""Procedural, seeded inputs for tests and bench.py: meshes, sensor models, poses,
particle clouds.  Pure input generation -- no hot-path arithmetic lives here.
"""
import math

import numpy as np

from .types import PARTICLE_ATTRIBUTES, TRANSFORM, euler_to_quat, spherical_model, transform_from_rpy


def find_abc(C):
    """Factorisation rule of the reference's benchmark (lidar_corrector_embree_benchmark.cpp:17-36)."""
    A = int(math.sqrt(C))
    B = A
    res = A * B
    while res != C and B > 0 and A <= C:
        if res > C:
            B -= 1
        else:
            A += 1
        res = A * B
    if not (B > 0 and A <= C and res == C):
        raise ValueError("could not factorise %d" % C)
    return A, B


def uv_sphere(n_faces, radius=10.0):
    """UV sphere with exactly n_faces triangles (A x B quads, 2 triangles each; the two triangle fans at
    the poles contain zero-area triangles, which never intersect).  Normals point outward."""
    if n_faces % 2:
        raise ValueError("n_faces must be even")
    A, B = find_abc(n_faces // 2)  # A columns (longitude), B rows (latitude)
    lon = (np.arange(A + 1, dtype=np.float64) / A) * 2.0 * np.pi
    lat = (np.arange(B + 1, dtype=np.float64) / B) * np.pi - np.pi / 2
    LON, LAT = np.meshgrid(lon, lat)  # (B+1, A+1)
    v = np.stack([radius * np.cos(LAT) * np.cos(LON), radius * np.cos(LAT) * np.sin(LON), radius * np.sin(LAT)], -1)
    verts = v.reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(B), np.arange(A), indexing="ij")
    p00 = (j * (A + 1) + i).ravel()
    p01 = p00 + 1
    p10 = p00 + (A + 1)
    p11 = p10 + 1
    faces = np.empty((2 * A * B, 3), dtype=np.uint32)
    faces[0::2] = np.stack([p00, p01, p11], -1)
    faces[1::2] = np.stack([p00, p11, p10], -1)
    return verts, faces


def cube_room(side=10.0, grid=9):
    """Axis-aligned cube room centred at the origin, each wall a grid x grid quad mesh:
    6 * grid^2 * 2 triangles (972 for grid=9), normals inward."""
    h = side / 2.0
    lin = np.linspace(-h, h, grid + 1)
    verts, faces = [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            base = len(verts)
            u_ax, v_ax = [(1, 2), (2, 0), (0, 1)][axis]
            for b in lin:
                for a in lin:
                    p = [0.0, 0.0, 0.0]
                    p[axis] = sign * h
                    p[u_ax] = a
                    p[v_ax] = b
                    verts.append(p)
            for jj in range(grid):
                for ii in range(grid):
                    p00 = base + jj * (grid + 1) + ii
                    p01, p10, p11 = p00 + 1, p00 + grid + 1, p00 + grid + 2
                    if sign > 0:  # inward normal = -axis
                        faces.append([p00, p11, p01])
                        faces.append([p00, p10, p11])
                    else:
                        faces.append([p00, p01, p11])
                        faces.append([p00, p11, p10])
    return np.asarray(verts, dtype=np.float32), np.asarray(faces, dtype=np.uint32)


def noisy_room(n_faces_target=100000, side=20.0, height=6.0, noise=0.02, seed=1234, boxes=6):
    """Room with interior boxes and seeded vertex noise: a less symmetric mesh than the sphere, with
    occlusion and rays that miss (open ceiling).  Returns about n_faces_target triangles."""
    rng = np.random.RandomState(seed)
    quads = []  # (origin, u, v) rectangles
    hx, hz = side / 2.0, height
    quads.append((np.array([-hx, -hx, 0.0]), np.array([side, 0, 0.0]), np.array([0, side, 0.0])))  # floor
    quads.append((np.array([-hx, -hx, 0.0]), np.array([side, 0, 0.0]), np.array([0, 0, hz])))
    quads.append((np.array([-hx, hx, 0.0]), np.array([side, 0, 0.0]), np.array([0, 0, hz])))
    quads.append((np.array([-hx, -hx, 0.0]), np.array([0, side, 0.0]), np.array([0, 0, hz])))
    quads.append((np.array([hx, -hx, 0.0]), np.array([0, side, 0.0]), np.array([0, 0, hz])))
    for _ in range(boxes):
        c = rng.uniform(-hx * 0.8, hx * 0.8, size=2)
        s = rng.uniform(0.5, 2.0, size=3)
        o = np.array([c[0] - s[0] / 2, c[1] - s[1] / 2, 0.0])
        ex, ey, ez = np.array([s[0], 0, 0.0]), np.array([0, s[1], 0.0]), np.array([0, 0, s[2]])
        quads += [(o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez), (o + ez, ex, ey)]
    area = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in quads])
    per_area = n_faces_target / 2.0 / area.sum()
    verts, faces = [], []
    for (o, u, v), a in zip(quads, area):
        lu, lv = np.linalg.norm(u), np.linalg.norm(v)
        nu = max(1, int(round(math.sqrt(a * per_area) * math.sqrt(lu / lv))))
        nv = max(1, int(round(a * per_area / nu)))
        base = len(verts)
        for jj in range(nv + 1):
            for ii in range(nu + 1):
                verts.append(o + u * (ii / nu) + v * (jj / nv))
        for jj in range(nv):
            for ii in range(nu):
                p00 = base + jj * (nu + 1) + ii
                p01, p10, p11 = p00 + 1, p00 + nu + 1, p00 + nu + 2
                faces.append([p00, p01, p11])
                faces.append([p00, p11, p10])
    verts = np.asarray(verts, dtype=np.float64)
    verts += rng.uniform(-noise, noise, size=verts.shape)
    return verts.astype(np.float32), np.asarray(faces, dtype=np.uint32)


def nested_triangles(n, ratio, smallest):
    """n coaxial triangles of geometrically growing size stacked 1 cm apart: the SAH builder peels them off in thin groups, which
    makes deep trees out of a few hundred triangles"""
    k = np.arange(n, dtype=np.float64)
    s = smallest * ratio ** k
    z = 0.01 * k
    v = np.stack([np.stack([-s, -s, z], -1), np.stack([s, -s, z], -1), np.stack([np.zeros(n), s, z], -1)], 1).reshape(-1, 3)
    return v.astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def exp_chain(n, ratio=2.0, k0=None):
    """n triangles whose position AND size grow geometrically along x (sizes from ratio^k0 to ratio^(k0+n-1)): the cheapest SAH split
    always separates the few largest from all the rest, i.e. the binary tree wants to be a chain about n / 4 deep -- far beyond what
    a 64-entry traversal stack serves.  The builder's height budget (bvh_build.cpp: kMaxHeight2) must step in."""
    k0 = -(n // 2) if k0 is None else k0
    s = float(ratio) ** (k0 + np.arange(n, dtype=np.float64))
    z = np.zeros(n)
    v = np.stack([np.stack([s, z, z], -1), np.stack([1.5 * s, 0.1 * s, z], -1), np.stack([s, 0.1 * s, 0.1 * s], -1)], 1).reshape(-1, 3)
    return v.astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def cad_mix(n_detail=100000, room=40.0, beam_yaw_deg=0.0, beam_tilt_deg=0.0, n_beams=8):
    """what a CAD export next to scanned detail looks like: a 40 m hall of TWELVE triangles (two per wall), eight 30 m beams of twelve
    triangles each crossing it, and a finely tessellated object (a UV sphere of radius 3) in the middle -- a few hundred-metre-scale
    triangles among 10^5 centimetre-scale ones.  beam_yaw_deg / beam_tilt_deg turn the beams out of the axes (about z, then about the
    turned y): long thin DIAGONAL triangles, whose boxes cover most of the hall -- the case spatial splits are for."""
    h = room / 2.0
    vs, fs = [], []
    first_beam_vertex = [None]

    def box(lo, hi):
        base = len(vs)
        for z in (lo[2], hi[2]):
            for y in (lo[1], hi[1]):
                for x in (lo[0], hi[0]):
                    vs.append([x, y, z])
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        for a, b, c, d in quads:
            fs.append([base + a, base + b, base + c])
            fs.append([base + a, base + c, base + d])

    box((-h, -h, -3.5), (h, h, 8.0))
    first_beam_vertex[0] = len(vs)
    for k in range(n_beams):
        y = -14.0 + 28.0 * k / max(n_beams - 1, 1)
        box((-15.0, y, 5.0 + 0.1 * (k % 8)), (15.0, y + 0.2, 5.3 + 0.1 * (k % 8)))
    if beam_yaw_deg != 0.0 or beam_tilt_deg != 0.0:
        cz, sz = math.cos(math.radians(beam_yaw_deg)), math.sin(math.radians(beam_yaw_deg))
        cy, sy = math.cos(math.radians(beam_tilt_deg)), math.sin(math.radians(beam_tilt_deg))
        Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
        Ry = np.array([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
        R = Rz @ Ry
        c = np.array([0.0, 0.0, 5.0])
        for i in range(first_beam_vertex[0], len(vs)):
            vs[i] = (R @ (np.asarray(vs[i]) - c) + c).tolist()
    sv, sf = uv_sphere(n_detail, radius=3.0)
    base = len(vs)
    verts = np.concatenate([np.asarray(vs, np.float32), sv])
    faces = np.concatenate([np.asarray(fs, np.uint32), sf + np.uint32(base)])
    return verts, faces


def sliver_fan(n, radius=10.0):
    """n long thin triangles that all share the apex at the origin (a disc cut like a pie, rim height wobbling): every triangle's box
    reaches the centre, the boxes overlap massively -- the classic bad case of an object-split BVH"""
    a = np.linspace(0.0, 2.0 * np.pi, n + 1)
    rim = np.stack([radius * np.cos(a), radius * np.sin(a), 0.3 * np.sin(7.0 * a)], -1)
    v = np.concatenate([np.zeros((1, 3)), rim]).astype(np.float32)
    f = np.stack([np.zeros(n, np.uint32), 1 + np.arange(n, dtype=np.uint32), 2 + np.arange(n, dtype=np.uint32)], -1)
    return v, f.astype(np.uint32)


# ---- sensor models -----------------------------------------------------------
def model_c1():
    """32x32, phi in [-45,45] deg, theta full circle, range [0.1, 100]."""
    f = np.float32
    return spherical_model(f(-math.pi / 4), f((math.pi / 2) / 31), 32, f(-math.pi), f(2 * math.pi / 32), 32, f(0.1), f(100.0))


def model_c2():
    """128x1024, phi in [-22.5,22.5] deg, theta full circle, range [0.3, 120]."""
    f = np.float32
    return spherical_model(f(-math.pi / 8), f((math.pi / 4) / 127), 128, f(-math.pi), f(2 * math.pi / 1024), 1024,
                           f(0.3), f(120.0))


def model_vlp16_900(range_min=0.0):
    """rmagine vlp16_900(): 16x900, phi in [-15,15] deg (lidar_corrector_embree_benchmark.cpp:91-93 sets range.min=0)."""
    f = np.float32
    return spherical_model(f(-15.0 * math.pi / 180), f((30.0 * math.pi / 180) / 15), 16, f(-math.pi), f(2 * math.pi / 900),
                           900, f(range_min), f(130.0))


def model_pf16(range_min=0.05, range_max=80.0):
    """16x16 particle-filter sensor: phi in [-15,15] deg, theta full circle."""
    f = np.float32
    return spherical_model(f(-15.0 * math.pi / 180), f((30.0 * math.pi / 180) / 15), 16, f(-math.pi), f(2 * math.pi / 16),
                           16, f(range_min), f(range_max))


def model_directions(model):
    """(H*W, 3) float32 directions of a spherical model in buffer order (vid*W + hid); used to turn a
    spherical scan into an O1Dn model / PF beams.  float32 trig of float32 angles, like the kernels' tables."""
    H, W = model.phi.size, model.theta.size
    phi = (np.float32(model.phi.min) + np.arange(H, dtype=np.float32) * np.float32(model.phi.inc)).astype(np.float32)
    th = (np.float32(model.theta.min) + np.arange(W, dtype=np.float32) * np.float32(model.theta.inc)).astype(np.float32)
    cp, sp = np.cos(phi).astype(np.float32), np.sin(phi).astype(np.float32)
    ct, st = np.cos(th).astype(np.float32), np.sin(th).astype(np.float32)
    d = np.empty((H, W, 3), dtype=np.float32)
    d[..., 0] = cp[:, None] * ct[None, :]
    d[..., 1] = cp[:, None] * st[None, :]
    d[..., 2] = sp[:, None] * np.ones((1, W), np.float32)
    return d.reshape(-1, 3)


# ---- poses -----------------------------------------------------------------------------------------
def pose_c2_truth():
    s = math.sqrt(2.0)
    return transform_from_rpy((0.37 * s, -0.21 * s, 0.13 * s), (0.02, -0.03, 0.4))


def pose_c2_perturbation():
    return transform_from_rpy((0.2, 0.1, 0.05), (0.0, 0.0, 2.0 * math.pi / 180))


def tsb_offset():
    """the one non-identity Tsb of SURVEY.md 8(d): t=(0.1,0,0.3), yaw 10 deg"""
    return transform_from_rpy((0.1, 0.0, 0.3), (0.0, 0.0, 10.0 * math.pi / 180))


def uniform_particles(n, seed=42, bb_min=(-8, -8, -1, 0, 0, -math.pi), bb_max=(8, 8, 1, 0, 0, math.pi)):
    """RmclNode::initSamplesUniform (rmcl_localization.cpp:277-342): uniform poses in a 6-D box,
    attrs = Gaussian1D::Identity() with mean 1."""
    rng = np.random.RandomState(seed)
    lo, hi = np.asarray(bb_min, dtype=np.float64), np.asarray(bb_max, dtype=np.float64)
    u = rng.uniform(size=(n, 6)) * (hi - lo) + lo
    poses = np.zeros(n, dtype=TRANSFORM)
    cr, sr = np.cos(u[:, 3] / 2), np.sin(u[:, 3] / 2)
    cp, sp = np.cos(u[:, 4] / 2), np.sin(u[:, 4] / 2)
    cy, sy = np.cos(u[:, 5] / 2), np.sin(u[:, 5] / 2)
    poses["R"]["x"] = sr * cp * cy - cr * sp * sy
    poses["R"]["y"] = cr * sp * cy + sr * cp * sy
    poses["R"]["z"] = cr * cp * sy - sr * sp * cy
    poses["R"]["w"] = cr * cp * cy + sr * sp * sy
    poses["t"]["x"], poses["t"]["y"], poses["t"]["z"] = u[:, 0], u[:, 1], u[:, 2]
    attrs = np.zeros(n, dtype=PARTICLE_ATTRIBUTES)
    attrs["likelihood"]["mean"] = 1.0
    return poses, attrs


def converged_particles(n, centre, sigma_t=0.25, sigma_yaw_deg=5.0, seed=42):
    """a converged particle cloud: poses ~ centre o N(0, sigma_t) in x / y (z: sigma_t / 4), yaw ~ N(0, sigma_yaw), roll = pitch = 0 in
    the centre's frame -- the filter's steady state (the reference resamples around the mode: GladiatorResamplerGPU noise terms);
    attrs as uniform_particles.  Seeded, order random (as a tournament leaves it)."""
    from . import types as T
    rng = np.random.RandomState(seed)
    d = np.zeros(n, dtype=TRANSFORM)
    yaw = rng.normal(0.0, math.radians(sigma_yaw_deg), n)
    d["R"]["z"], d["R"]["w"] = np.sin(yaw / 2), np.cos(yaw / 2)
    d["t"]["x"], d["t"]["y"], d["t"]["z"] = rng.normal(0, sigma_t, n), rng.normal(0, sigma_t, n), rng.normal(0, sigma_t / 4, n)
    # pose_i = centre * d_i, with the library's own arithmetic (vectorised restatement of Transform::operator*)
    c = np.ascontiguousarray(centre, dtype=TRANSFORM).reshape(1)[0]
    cq = np.array([c["R"][k] for k in "xyzw"], np.float64)
    ct = np.array([c["t"][k] for k in "xyz"], np.float64)
    dq = np.stack([d["R"][k].astype(np.float64) for k in "xyzw"], -1)
    dt = np.stack([d["t"][k].astype(np.float64) for k in "xyz"], -1)

    def qmul(a, b):
        ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
        bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
        return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)

    q = qmul(cq[None, :], dq)
    pv = np.concatenate([dt, np.zeros((n, 1))], -1)
    cinv = cq * np.array([-1, -1, -1, 1.0])
    rt = qmul(qmul(cq[None, :], pv), cinv[None, :])[:, :3] + ct
    poses = np.zeros(n, dtype=TRANSFORM)
    for i, k in enumerate("xyzw"):
        poses["R"][k] = q[:, i]
    for i, k in enumerate("xyz"):
        poses["t"][k] = rt[:, i]
    attrs = np.zeros(n, dtype=PARTICLE_ATTRIBUTES)
    attrs["likelihood"]["mean"] = 1.0
    _ = T
    return poses, attrs


def morton_order_xy_yaw(poses, bits=10):
    """slot -> particle index sorted by a Morton key of (x, y, yaw), `bits` bits each over the cloud's own bounding box (the slot order
    rmclhip_pf_set_mapping accepts; a host form: the mapping measured neutral, profiles/r04_pf_converged_mapping.txt, so no device sort was built)"""
    x, y = poses["t"]["x"].astype(np.float64), poses["t"]["y"].astype(np.float64)
    yaw = 2.0 * np.arctan2(poses["R"]["z"].astype(np.float64), poses["R"]["w"].astype(np.float64))

    def quant(v):
        lo, hi = v.min(), v.max()
        s = (2 ** bits - 1) / (hi - lo) if hi > lo else 0.0
        return np.clip(((v - lo) * s), 0, 2 ** bits - 1).astype(np.uint64)

    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out

    key = spread(quant(yaw)) << np.uint64(2) | spread(quant(x)) << np.uint64(1) | spread(quant(y))
    return np.argsort(key, kind="stable").astype(np.uint32)


__all__ = [n for n in dir() if not n.startswith("_")]
_ = euler_to_quat  # re-exported for callers
