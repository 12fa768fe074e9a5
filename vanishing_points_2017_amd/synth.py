"""Seeded synthetic line / CNN-response / sphere-raster workloads.

The reference's datasets (YUD, ECD, HLW) and trained weights are not available
offline, so every BASELINE.json config is restated as seeded synthetic inputs
with the dataset's *shape*.  One "scene" mirrors the reference's per-image datum
(`evaluation.py:152-177`): line segments ``lp`` (N x 4, normalised image
coordinates, y up), homogeneous lines ``l = cross(p1, p2)`` (`evaluation.py:161-168`),
a 20 x 20 float32 ``cnn_response`` laid out like the `sigout` blob
(`evaluation.py:34-38`; row = beta bin ascending, column = alpha bin ascending).  A scene carries NO
raster (``sphere_image`` is None): the reference makes the 500 x 500 uint8 raster from the lines
(`evaluation.py:175`, `sphere_mapping.py:36-72`), and so does everything here -- the product through
``vpk_sphere_raster`` (`sphere_mapping.attach_rasters`), the fixtures through the reference's own
`sphere_line_plot` (the fixture generator, build container only), the CPU tests through the test suite's own restatement.

Seeds follow SURVEY 8d: ``seed = 1000 * config_id + image_index`` with
``numpy.random.RandomState``.
"""
import numpy as np

CONFIGS = {
    # id: (name, n_images, (n_lo, n_hi), (vp_lo, vp_hi), aspect_h)
    1: ("single-example", 1, (800, 800), (3, 3), 0.667),
    2: ("yud-shape", 102, (100, 400), (3, 3), 0.75),
    3: ("ecd-shape", 103, (300, 1200), (3, 8), 0.75),
    4: ("hlw-shape", 2018, (100, 1000), (3, 5), 0.75),
    5: ("stress", 10000, (1000, 1000), (8, 8), 0.75),
}


def _rotation(rs, mode="euler"):
    if mode == "qr":
        q, r = np.linalg.qr(rs.randn(3, 3))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        return q
    yaw = rs.uniform(0, 2 * np.pi)
    pitch = rs.normal(0, 0.12)
    roll = rs.normal(0, 0.06)
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    r_yaw = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    r_pitch = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    r_roll = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return r_roll.dot(r_pitch).dot(r_yaw)


def vp_to_cell(vp, grid=20):
    """Grid cell (row=beta bin, col=alpha bin) of a unit VP direction.

    Inverse of `coordinate_conversion.py:53-61` + `:23-35` on a grid x grid map."""
    vp = vp / np.linalg.norm(vp)
    if vp[2] < 0:
        vp = -vp
    beta = np.arcsin(np.clip(vp[1], -1, 1))
    alpha = np.arcsin(np.clip(vp[0] / max(np.cos(beta), 1e-12), -1, 1))
    col = int(np.clip(np.rint((alpha / np.pi + 0.5 - 0.5 / grid) * grid), 0, grid - 1))
    row = int(np.clip(np.rint((beta / np.pi + 0.5 - 0.5 / grid) * grid), 0, grid - 1))
    return row, col


def make_scene(seed, n_lines, n_vps=3, focal=2.1, aspect_h=0.75, outlier_frac=0.25,
               noise=0.004, rot_mode="euler", raster=None, size=500):
    """One synthetic image datum.  Returns a dict with l, lp, cnn_response, sphere_image (None unless a
    ``raster`` callable (lines, size, alpha) -> uint8 image is given), true_vps (unit, z >= 0), true_horizon
    (homogeneous line)."""
    rs = np.random.RandomState(seed)
    K = np.diag([focal, focal, 1.0])
    R = _rotation(rs, rot_mode)
    dirs = [R[:, 0], R[:, 1], R[:, 2]]
    # extra horizontal directions (ECD-shape): rotate x about the vertical axis R[:,1]
    for _ in range(max(0, n_vps - 3)):
        t = rs.uniform(0.2, np.pi - 0.2)
        dirs.append(np.cos(t) * R[:, 0] + np.sin(t) * R[:, 2])
    vps = []
    for d in dirs[:n_vps]:
        v = K.dot(d)
        v = v / np.linalg.norm(v)
        if v[2] < 0:
            v = -v
        vps.append(v)
    vps = np.array(vps)
    hor = np.linalg.inv(K).T.dot(R[:, 1])  # horizon line = K^-T * vertical direction
    hor = hor / np.linalg.norm(hor[0:2])

    mids = np.stack([rs.uniform(-0.9, 0.9, n_lines), rs.uniform(-0.9, 0.9, n_lines) * aspect_h], 1)
    which = rs.randint(0, n_vps, n_lines)
    is_out = rs.uniform(size=n_lines) < outlier_frac
    length = rs.uniform(0.03, 0.4, n_lines)
    v = vps[which]
    d = np.stack([v[:, 0] - mids[:, 0] * v[:, 2], v[:, 1] - mids[:, 1] * v[:, 2]], 1)
    rnd = rs.uniform(0, np.pi, n_lines)
    d[is_out] = np.stack([np.cos(rnd), np.sin(rnd)], 1)[is_out]
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    p1 = mids + 0.5 * length[:, None] * d + rs.normal(0, noise, (n_lines, 2))
    p2 = mids - 0.5 * length[:, None] * d + rs.normal(0, noise, (n_lines, 2))
    lp = np.ascontiguousarray(np.concatenate([p1, p2], 1))
    h1 = np.concatenate([p1, np.ones((n_lines, 1))], 1)
    h2 = np.concatenate([p2, np.ones((n_lines, 1))], 1)
    l = np.ascontiguousarray(np.cross(h1, h2))

    cnn = rs.uniform(0, 0.08, (20, 20))
    for vp in vps:
        r, c = vp_to_cell(vp)
        cnn[r, c] = rs.uniform(0.5, 0.95)
    for _ in range(6):
        cnn[rs.randint(0, 20), rs.randint(0, 20)] = rs.uniform(0.3, 0.6)
    cnn = cnn.astype(np.float32)

    sphere = raster(l.copy(), size, 0.1) if raster is not None else None
    assoc = np.where(is_out, -1, which)
    return {"l": l, "lp": lp, "cnn_response": cnn, "sphere_image": sphere, "true_vps": vps,
            "true_horizon": hor, "true_assoc": assoc, "seed": seed,
            "image_shape": (int(round(640 * aspect_h)), 640)}


def config_scenes(config_id, count=None, start=0, raster=None):
    """Generator over the scenes of one BASELINE.json config."""
    name, n_img, (n_lo, n_hi), (v_lo, v_hi), aspect_h = CONFIGS[config_id]
    n_img = n_img if count is None else min(n_img, count)
    for idx in range(start, start + n_img):
        seed = 1000 * config_id + idx
        rs = np.random.RandomState(seed + 7919)
        n = int(rs.randint(n_lo, n_hi + 1))
        nv = int(rs.randint(v_lo, v_hi + 1))
        yield make_scene(seed, n, nv, aspect_h=aspect_h, raster=raster)


def stress_init_vps(seed, m=8):
    """init_vp (m x 3) for the stress unit: 8 seeded directions, z >= 0."""
    rs = np.random.RandomState(seed + 104729)
    v = rs.randn(m, 3)
    v[:, 2] = np.abs(v[:, 2]) + 0.2
    return v / np.linalg.norm(v, axis=1, keepdims=True)
