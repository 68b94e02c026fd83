"""Seeded synthetic stand-ins for the Blender-Lego inputs of the Instant-NGP path (there is no
network for datasets): cameras, "Lego-shaped" occupancy, rays, tables and weights.

Shapes follow the reference: poses are the python [n_img, 4, 3] matrices that
`poses_nerf2ngp` (/root/reference/xrnerf/datasets/utils/hashnerf.py:4-23) produces, metadata is
the 11-float TrainingImageMetadata row of `HashNerfDataset.get_alldata`
(/root/reference/xrnerf/datasets/hashnerf_dataset.py:56-73), the density grid is the
[8*128^3] Morton-ordered cascade of `NGPGridSampler` (samplers/ngp_grid_sampler.py:47-69).
"""
import numpy as np

GRID = 128
G3 = GRID ** 3
CASCADES = 8
LEGO_FOCAL = 0.5 * 800 / np.tan(0.5 * 0.6911112070083618)  # 1111.11 (transforms_*.json camera_angle_x)


def expand_bits(v):
    v = np.asarray(v, dtype=np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z):
    return expand_bits(x) | (expand_bits(y) << np.uint32(1)) | (expand_bits(z) << np.uint32(2))


def morton3d_invert(x):
    x = np.asarray(x, dtype=np.uint32) & np.uint32(0x49249249)
    x = (x | (x >> np.uint32(2))) & np.uint32(0xc30c30c3)
    x = (x | (x >> np.uint32(4))) & np.uint32(0x0f00f00f)
    x = (x | (x >> np.uint32(8))) & np.uint32(0xff0000ff)
    x = (x | (x >> np.uint32(16))) & np.uint32(0x0000ffff)
    return x


def blender_pose(theta_deg, phi_deg, radius):
    """camera-to-world 4x4 in the NeRF-Blender convention (camera looks down -z at the origin)."""
    t = np.eye(4); t[2, 3] = radius
    ph = np.deg2rad(phi_deg)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    th = np.deg2rad(theta_deg)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = rt @ rp @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w
    return c2w


def poses_nerf2ngp(poses44, scale=0.33, offset=0.5):
    """same maths as the reference's poses_nerf2ngp (correct_pose [1,-1,-1], cycle [1,2,0]); returns
    [n, 4, 3] float32."""
    out = []
    for p in np.asarray(poses44, dtype=np.float64):
        m = p[:3, :].copy()
        m[:, 1] *= -1
        m[:, 2] *= -1
        m[:, 3] = m[:, 3] * scale + offset
        m = m[[1, 2, 0]]
        out.append(m)
    return np.array(out).astype(np.float32).transpose(0, 2, 1).copy()


def lego_cameras(n_img=100, seed=1, radius=4.0):
    """n_img cameras on the upper Blender hemisphere, converted to NGP space -> [n,4,3] f32."""
    rng = np.random.default_rng(seed)
    th = rng.uniform(-180, 180, n_img)
    ph = -np.rad2deg(np.arccos(rng.uniform(0.0, 0.95, n_img)))  # elevation above the table
    return poses_nerf2ngp(np.stack([blender_pose(t, p, radius) for t, p in zip(th, ph)]))


def metadata_rows(n_img, focal):
    row = np.array([0, 0, 0, 0, 0.5, 0.5, focal, focal, 0, 0, 0], np.float32)
    return np.repeat(row[None], n_img, 0).copy()


def lego_density_grid(seed=2, fill=0.07, value=1.0, n_boxes=40):
    """Union of axis-aligned boxes centred near 0.5 filling ~`fill` of the level-0 cells, rasterised
    into all 8 cascades in Morton order. Occupied cells get `value`, empty cells 0."""
    rng = np.random.default_rng(seed)
    lo_all, hi_all = [], []
    # a chassis + random bricks, rejection-tuned to the requested fill fraction
    lo_all.append(np.array([0.30, 0.38, 0.33])); hi_all.append(np.array([0.72, 0.62, 0.47]))
    vol = np.prod(hi_all[0] - lo_all[0])
    while vol < fill and len(lo_all) < n_boxes:
        c = rng.uniform(0.28, 0.72, 3)
        h = rng.uniform(0.02, 0.09, 3)
        lo_all.append(c - h); hi_all.append(c + h)
        vol += np.prod(2 * h) * 0.6
    lo_all, hi_all = np.array(lo_all), np.array(hi_all)
    ax = np.arange(GRID, dtype=np.float32)
    xi, yi, zi = np.meshgrid(np.arange(GRID, dtype=np.uint32), np.arange(GRID, dtype=np.uint32),
                             np.arange(GRID, dtype=np.uint32), indexing='ij')
    mort = morton3d(xi, yi, zi).ravel()
    grid = np.zeros(CASCADES * G3, np.float32)
    for level in range(CASCADES):
        pos = ((ax + 0.5) / GRID - 0.5) * (2.0 ** level) + 0.5
        half = 0.5 * (2.0 ** level) / GRID
        # per-axis interval overlap is separable: occ[x,y,z] = any_b in_x[b,x] & in_y[b,y] & in_z[b,z]
        ins = [((pos[None, :] + half >= lo_all[:, a:a + 1]) & (pos[None, :] - half <= hi_all[:, a:a + 1]))
               .astype(np.float32) for a in range(3)]
        occ = np.einsum('bx,by,bz->xyz', ins[0], ins[1], ins[2]) > 0
        lvl = np.zeros(G3, np.float32)
        lvl[mort] = np.where(occ.ravel(), value, 0.0)
        grid[level * G3:(level + 1) * G3] = lvl
    return grid


def sphere_density_grid(radius=0.3, value=1.0):
    idx = np.arange(G3, dtype=np.uint32)
    xyz = np.stack([morton3d_invert(idx >> np.uint32(k)) for k in range(3)], -1).astype(np.float32)
    grid = np.zeros(CASCADES * G3, np.float32)
    for level in range(CASCADES):
        pos = ((xyz + 0.5) / GRID - 0.5) * (2.0 ** level) + 0.5
        grid[level * G3:(level + 1) * G3] = np.where(np.linalg.norm(pos - 0.5, axis=1) < radius, value, 0.0)
    return grid


def camera_rays(pose43, H, W, focal, pix=None):
    """fp32 ray generation with the maths of get_rays_np_hash
    (/root/reference/xrnerf/datasets/load_data/get_rays.py:35-69); `pix` = flat pixel indices."""
    pose43 = np.asarray(pose43, np.float32)
    if pix is None:
        pix = np.arange(H * W)
    j, i = np.divmod(np.asarray(pix), W)
    f = np.float32(focal)
    dx = (i.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * W)) / f
    dy = (j.astype(np.float32) + np.float32(0.5) - np.float32(0.5 * H)) / f
    dirs = np.stack([dx, dy, np.ones_like(dx)], -1)
    c2w = pose43.T
    d = (dirs[:, None, :] * c2w[None, :3, :3]).sum(-1).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(c2w[:3, 3], d.shape).astype(np.float32).copy()
    return o, d.astype(np.float32)


def training_rays(poses, n_rays, H=800, W=800, focal=LEGO_FOCAL, seed=7):
    """n_rays random (image, pixel) pairs -> rays_o, rays_d [n,3] f32, img_ids [n,1] i32."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, poses.shape[0], n_rays)
    pix = rng.integers(0, H * W, n_rays)
    o = np.zeros((n_rays, 3), np.float32)
    d = np.zeros((n_rays, 3), np.float32)
    for k in np.unique(img):
        m = img == k
        o[m], d[m] = camera_rays(poses[k], H, W, focal, pix[m])
    return o, d, img.astype(np.int32)[:, None]


def hash_table(n_params, seed=3, scale=1e-4):
    """tcnn initialises grids U(-1e-4, 1e-4)."""
    return np.random.default_rng(seed).uniform(-scale, scale, n_params).astype(np.float32)


def mlp_weights(in_pad, width, n_hidden, out_pad, seed=4):
    """Xavier-uniform per layer, flat row-major [out,in] matrices in layer order."""
    rng = np.random.default_rng(seed)
    dims = [in_pad] + [width] * n_hidden + [out_pad]
    ws = []
    for a, b in zip(dims[:-1], dims[1:]):
        lim = np.sqrt(6.0 / (a + b))
        ws.append(rng.uniform(-lim, lim, (b, a)).astype(np.float32).ravel())
    return np.concatenate(ws)
