"""Seeded random scenes for differential tests of the HIP path against the CPU oracle.

Every seed draws its own mix: spheres and triangles (flat and with vertex normals) in clusters of
very different size, a length scale between 1e-3 and 1e3 and an offset of the whole scene from
the origin of up to a thousand times its size (the box tests of the hierarchies must stay
conservative at every magnitude), mirrors, glass and both at once, all four kinds of light with
spot angles on either side of the fall-off, perspective or orthographic camera, odd film sizes."""
import numpy as np

import raytracer_amd as ra


def _unit(v):
    n = np.linalg.norm(v)
    return v / n if n > 0 else np.array([0.0, 0.0, 1.0])


def random_scene(seed, big=False):
    """big: thousands of primitives and films of up to 256 x 160 (deep hierarchies, many packets; seconds of oracle time)"""
    rng = np.random.default_rng((1000 if not big else 500000) + seed)
    scale = float(10.0 ** rng.choice([-3, -1, 0, 0, 1, 3]))
    offset = rng.normal(size=3) * scale * float(rng.choice([0, 0, 1, 30, 1000]))
    n_sph = int(rng.choice([0, 1, 5, 40, 200] if not big else [0, 300, 2000, 6000]))
    n_tri = int(rng.choice([0, 1, 12, 100, 400] if not big else [0, 500, 5000, 20000]))
    if n_sph + n_tri == 0:
        n_sph = 3
    n_mat = int(rng.integers(1, 9))
    mats = np.zeros((n_mat, 14), np.float32)
    mats[:, 0:3] = rng.random((n_mat, 3)) * 0.6
    mats[:, 3:6] = rng.random((n_mat, 3))
    mats[:, 9] = rng.choice([1, 5, 20, 80], n_mat)
    mats[:, 13] = 1
    for m in range(n_mat):
        kind = rng.integers(0, 4)                     # matte, mirror, glass, both
        if kind in (1, 3):
            mats[m, 6:9] = rng.random(3) * 0.8
        if kind in (2, 3):
            mats[m, 10:13] = rng.random(3) * 0.9
            mats[m, 13] = rng.choice([1.0, 1.1, 1.5, 2.4, 0.7])
    n_clusters = int(rng.integers(1, 5))
    centres = rng.normal(size=(n_clusters, 3)) * 3.0 + np.array([0, 0, 8.0])
    spread = 10.0 ** rng.uniform(-1.5, 0.6, n_clusters)

    def points(n):
        c = rng.integers(0, n_clusters, n)
        return centres[c] + rng.normal(size=(n, 3)) * spread[c, None], spread[c]

    sph = np.zeros((n_sph, 4), np.float32)
    if n_sph:
        p, s = points(n_sph)
        sph[:, 0:3] = p
        sph[:, 3] = s * 10.0 ** rng.uniform(-1.5, 0.0, n_sph)
    tv = np.zeros((n_tri, 9), np.float32)
    tn = np.zeros((n_tri, 9), np.float32)
    tf = np.zeros(n_tri, np.int32)
    if n_tri:
        p, s = points(n_tri)
        size = s * 10.0 ** rng.uniform(-1.5, 0.3, n_tri)
        for k in range(3):
            tv[:, 3 * k:3 * k + 3] = p + rng.normal(size=(n_tri, 3)) * size[:, None]
        axis = rng.random(n_tri) < 0.3              # axis-aligned ones: boxes with no thickness
        ax = rng.integers(0, 3, n_tri)
        for k in range(1, 3):
            tv[axis, 3 * k + ax[axis]] = tv[axis, ax[axis]]
        tf[:] = rng.random(n_tri) < 0.4
        e1, e2 = tv[:, 3:6] - tv[:, 0:3], tv[:, 6:9] - tv[:, 0:3]
        fn = np.cross(e1, e2)
        fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
        for k in range(3):
            nk = fn + rng.normal(size=(n_tri, 3)) * 0.3
            tn[:, 3 * k:3 * k + 3] = nk / np.maximum(np.linalg.norm(nk, axis=1, keepdims=True), 1e-30)
    if rng.random() < 0.5:                           # a floor under everything
        y = float(centres[:, 1].min() - 3 * spread.max())
        q = np.array([[-40, y, -10, 40, y, -10, 40, y, 60], [-40, y, -10, 40, y, 60, -40, y, 60]], np.float32)
        tv = np.concatenate([tv, q]); tn = np.concatenate([tn, np.tile([[0, 1, 0] * 3], (2, 1)).astype(np.float32)])
        tf = np.concatenate([tf, np.zeros(2, np.int32)])
        n_tri += 2
    lights = []
    for _ in range(int(rng.integers(0, 5))):
        kind = int(rng.integers(1, 4))
        pos = centres[rng.integers(0, n_clusters)] + rng.normal(size=3) * 6.0
        direction = _unit(centres[rng.integers(0, n_clusters)] - pos + rng.normal(size=3))
        col = rng.uniform(0.1, 0.7, 3) if kind == 1 else rng.uniform(5, 60, 3)
        a1 = float(rng.uniform(0.05, 1.2))
        a2 = a1 + float(rng.uniform(0.05, 0.8)) if rng.random() < 0.8 else float(rng.uniform(20, 90))
        lights.append([kind, *col, *pos, *direction, a1, a2])
    lights = np.array(lights, np.float32).reshape(-1, 12)
    target = centres[rng.integers(0, n_clusters)]
    cam_pos = target + _unit(rng.normal(size=3)) * float(rng.uniform(2, 14))
    cam_dir = _unit(target - cam_pos + rng.normal(size=3) * 0.3)
    up = rng.normal(size=3)
    up = _unit(up - cam_dir * float(np.dot(up, cam_dir)))
    # positions scale and move; directions, colours and angles do not
    S = np.float64(scale)
    sph64 = sph.astype(np.float64)
    sph64[:, 0:3] = sph64[:, 0:3] * S + offset
    sph64[:, 3] *= S
    tv64 = tv.astype(np.float64).reshape(-1, 3, 3) * S + offset
    if len(lights):
        l64 = lights.astype(np.float64)
        l64[:, 4:7] = l64[:, 4:7] * S + offset
        l64[:, 1:4] = np.where(l64[:, 0:1] == 1, l64[:, 1:4], l64[:, 1:4] * S * S)   # inverse-square lights keep their brightness
        lights = l64.astype(np.float32)
    proj = ra.RT_PROJ_PERSP if rng.random() < 0.8 else 1 - ra.RT_PROJ_PERSP
    width, height = int(rng.integers(1, 97 if not big else 257)), int(rng.integers(1, 65 if not big else 161))
    spp = int(rng.choice([1, 1, 2, 3, 5]))
    max_depth = int(rng.choice([0, 1, 2, 5, 5, 7]))
    jitter_seed = int(rng.integers(1, 1 << 30))
    # rarer cases, for seeds from 20000 on (the scenes of lower seeds stay what they were when findings were
    # recorded under their numbers), drawn from a stream of their own
    odd = np.random.default_rng(7000 + seed)
    rare = (lambda p: odd.random() < p) if seed >= 20000 else (lambda p: False)
    cam64 = cam_pos * S + offset
    if rare(0.1) and n_sph:                       # spheres of radius zero and below
        pick = odd.random(n_sph) < 0.3
        sph64[pick, 3] *= odd.choice([0.0, -1.0], int(pick.sum()))
    if rare(0.1) and n_sph:                                # the eye inside a sphere
        k = int(np.argmax(np.abs(sph64[:, 3])))
        cam64 = sph64[k, 0:3] + odd.normal(size=3) * 0.2 * abs(sph64[k, 3])
    if rare(0.12):                                         # long chains of mirror bounces: many thin generations
        max_depth = int(odd.choice([12, 25]))
        mats[:, 6:9] = odd.uniform(0.3, 0.9, (n_mat, 3))
        mats[:, 10:13] = 0
    if rare(0.1):                                          # many lights: one shadow query per hit and light
        extra = []
        for _ in range(int(odd.integers(5, 10))):
            kind = int(odd.integers(1, 4))
            pos = (centres[odd.integers(0, n_clusters)] + odd.normal(size=3) * 5.0) * S + offset
            col = odd.uniform(0.05, 0.3, 3) if kind == 1 else odd.uniform(2, 20, 3) * S * S
            extra.append([kind, *col, *pos, *_unit(odd.normal(size=3)), float(odd.uniform(0.1, 1.0)), float(odd.uniform(1.0, 1.6))])
        lights = np.concatenate([np.asarray(lights, np.float32).reshape(-1, 12), np.array(extra, np.float32)])
    return ra.SceneArrays(
        width=width, height=height, spp=spp, max_depth=max_depth, proj=int(proj), seed=jitter_seed,
        cam_pos=cam64.astype(np.float32), cam_dir=cam_dir.astype(np.float32), cam_up=up.astype(np.float32),
        cam_half_angle=float(np.float32(rng.uniform(0.15, 1.0))),
        background=rng.random(3).astype(np.float32), ambient=(rng.random(3) * 0.4).astype(np.float32),
        spheres=sph64.astype(np.float32), sphere_mat=rng.integers(0, n_mat, n_sph).astype(np.int32),
        tri_verts=tv64.reshape(-1, 9).astype(np.float32), tri_normals=tn, tri_flags=tf,
        tri_mat=rng.integers(0, n_mat, n_tri).astype(np.int32), materials=mats, lights=lights).normalise()


# how the walks of the hierarchies are forced (environment read when a context is created)
WALKS = {
    "packet": {"RT_PACKET_GENS": "99", "RT_PACKET_SHADOW_GENS": "99"},
    "lane": {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0"},
    "lane-wide": {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0", "RT_LANE_WIDE": "1"},
    "team": {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0", "RT_LANE_WIDE": "2"},
    # a grid of 8 blocks (20 waves): films of more than 1 300 rays walk in chunks drawn one after the other, as large frames do
    "lane-chunks": {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0", "RT_MAX_BLOCKS": "8"},
    "lane-wide-chunks": {"RT_PACKET_GENS": "0", "RT_PACKET_SHADOW_GENS": "0", "RT_LANE_WIDE": "1", "RT_MAX_BLOCKS": "8"},
}


def edge_grazing_case(trial, n=400_000, nt=400):
    """A sparse cloud of small triangles (every second trial: triangles lying in a coordinate plane, whose boxes have
    no thickness; every third: rays nearly in the triangle's plane) and n rays aimed within a few float roundings
    of an edge or a vertex of a random triangle, from origins up to thousands of units away -- the zone where the
    reference's float determinants decide by rounding.  Two far triangles stretch the region rays may start from.
    -> (scene, origins [n,3] float32, directions [n,3] float32)"""
    rng = np.random.default_rng(100 + trial)
    c = rng.uniform(-1, 1, (nt, 3))
    size = 10.0 ** rng.uniform(-3.0, -1.0, nt)
    v = c[:, None, :] + rng.normal(size=(nt, 3, 3)) * size[:, None, None]
    if trial % 2 == 1:
        ax = rng.integers(0, 3, nt)
        v[np.arange(nt), 1, ax] = v[np.arange(nt), 0, ax]
        v[np.arange(nt), 2, ax] = v[np.arange(nt), 0, ax]
    far = float(10.0 ** rng.uniform(1, 3.5))
    v[0] = np.array([[far, far, far], [far + 1, far, far], [far, far + 1, far]])
    v[1] = -v[0]
    tv = v.reshape(nt, 9).astype(np.float32)
    vf = tv.reshape(nt, 3, 3).astype(np.float64)
    fn = np.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-300)
    scene = ra.SceneArrays(width=8, height=8, tri_verts=tv, tri_normals=np.tile(fn, (1, 3)).astype(np.float32),
                           tri_flags=np.zeros(nt, np.int32), tri_mat=np.zeros(nt, np.int32),
                           cam_pos=np.array([0, 0, -5], np.float32)).normalise()
    k = rng.integers(2, nt, n)
    e = rng.integers(0, 3, n)
    a, b = vf[k, e], vf[k, (e + 1) % 3]
    s = np.where(rng.random(n) < 0.3, rng.integers(0, 2, n).astype(np.float64), rng.random(n))
    p = a + (b - a) * s[:, None]
    dist = 10.0 ** rng.uniform(0, np.log10(far), n)
    dirn = rng.normal(size=(n, 3))
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    if trial % 3 == 2:
        dirn -= fn[k] * (np.einsum("ij,ij->i", dirn, fn[k]) * rng.uniform(0.9, 1.0, n))[:, None]
        dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    o = np.clip(p - dirn * dist[:, None], -far, far + 1)       # inside the region the hierarchy is built for
    off = rng.normal(size=(n, 3))
    off -= dirn * np.einsum("ij,ij->i", off, dirn)[:, None]
    off /= np.maximum(np.linalg.norm(off, axis=1, keepdims=True), 1e-300)
    tgt = p + off * (rng.uniform(-4, 4, n) * 6e-8 * np.linalg.norm(p - o, axis=1))[:, None]
    d = (tgt - o) * rng.uniform(0.3, 2.0, (n, 1))
    return scene, o.astype(np.float32), d.astype(np.float32)
