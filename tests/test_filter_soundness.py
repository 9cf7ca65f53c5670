"""The FP32 candidate filter must never drop a (ray, sphere) pair the reference's f64
test accepts.  The device arithmetic is modelled in numpy float32 (tests/filter_model.py);
the f64 side is the reference's object-space quadratic (src/scene/shapes/spheres.rs:18-35
behind src/scene.rs:75-84), cross-checked against the CPU oracle on a subsample.

Adversarial inputs: for every (ray, sphere centre) pair the radius is tuned by bisection
to the grazing radius R* (discriminant = 0) and the filter is asked about R*(1 + delta)
for tiny positive delta, i.e. rays that only just touch.
"""
import numpy as np
import pytest

from filter_model import FilterScene


def exact_disc(p, d, dw, c, R):
    """Discriminant of the reference's quadratic for a sphere translate(c) * scale(R)."""
    R = np.asarray(R, np.float64)[..., None]
    dw = np.asarray(dw, np.float64)
    oo = (p - c) / R
    dd = (d - c * dw[..., None]) / R
    a = (dd * dd).sum(axis=-1) + dw * dw
    b = 2.0 * (oo * dd).sum(axis=-1)
    cc = (oo * oo).sum(axis=-1) - 1.0
    return b * b - 4.0 * a * cc


def grazing_radius(p, d, dw, c):
    """Smallest radius with a real root, by bisection on the exact discriminant."""
    lo = np.zeros(len(p)) + 1e-9
    hi = np.sqrt(((p - c) ** 2).sum(axis=-1)) * 1.5 + 1.0
    assert (exact_disc(p, d, dw, c, hi) >= 0).all()
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        hit = exact_disc(p, d, dw, c, mid) >= 0
        hi = np.where(hit, mid, hi)
        lo = np.where(hit, lo, mid)
    return hi


DELTAS = (1e-13, 1e-10, 1e-8, 1e-6, 1e-4, 1e-2, 0.5)


def _unit(v):
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


@pytest.mark.parametrize("scale", [1.0, 30.0, 300.0, 3000.0])
def test_primary_filter_keeps_every_grazing_hit(scale):
    rng = np.random.default_rng(int(scale))
    n = 20000
    origin = rng.normal(size=3) * scale * 0.3
    c = origin + _unit(rng.normal(size=(n, 3))) * rng.uniform(0.5, 1.0, size=(n, 1)) * scale * rng.uniform(0.05, 1, size=(n, 1))
    # aim near the limb of a ball of radius ~r0
    r0 = rng.uniform(0.05, 2.0, size=n)
    toward = c - origin
    perp = _unit(np.cross(toward, rng.normal(size=(n, 3))))
    d = _unit(toward + perp * r0[:, None])
    p = np.broadcast_to(origin, (n, 3))
    Rstar = grazing_radius(p, d, np.zeros(n), c)
    for delta in DELTAS:
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, np.zeros(n), c, R) >= 0
        fs = FilterScene(c, R)
        keep = fs.primary_pass(origin, d, pairwise=True)
        assert not (hit & ~keep).any(), f"dropped {int((hit & ~keep).sum())} grazing hits at delta={delta}"
        assert delta < 1e-6 or hit.mean() > 0.9  # below that the f64 discriminant itself is rounding noise


@pytest.mark.parametrize("light_w", [0.0, 1.0, 0.5])
@pytest.mark.parametrize("scale", [2.0, 50.0, 400.0])
def test_shadow_filter_keeps_every_grazing_hit(scale, light_w):
    rng = np.random.default_rng(int(scale * 10 + light_w * 3))
    n = 20000
    p = rng.normal(size=(n, 3)) * scale * 0.4
    L = np.array([0.3, 1.0, -0.2]) * scale + rng.normal(size=3)
    # sphere centres scattered around the segment p -> light and beyond
    s = rng.uniform(-0.5, 1.5, size=(n, 1))
    c = p + (L - p) * s + rng.normal(size=(n, 3)) * rng.uniform(0.01, 3.0, size=(n, 1))
    lv = L - p
    lw = light_w - 1.0
    m = np.sqrt((lv * lv).sum(axis=1) + lw * lw)
    d, dw = lv / m[:, None], lw / m
    Rstar = grazing_radius(p, d, dw, c)
    G = L - light_w * p
    for delta in DELTAS:
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, dw, c, R) >= 0
        fs = FilterScene(c, R)
        keep = fs.general_pass(p, G, np.full(n, lw), pairwise=True)
        assert not (hit & ~keep).any(), f"dropped {int((hit & ~keep).sum())} at delta={delta}"
        assert delta < 1e-6 or hit.mean() > 0.9  # below that the f64 discriminant itself is rounding noise


@pytest.mark.parametrize("scale", [2.0, 50.0, 400.0, 3000.0])
def test_tabulated_shadow_filter_keeps_every_grazing_hit(scale):
    """Lights with w = 0: the filter of the three-kernel path (tabulated K, expanded |u|^2)."""
    rng = np.random.default_rng(int(scale) + 5)
    n = 20000
    p = rng.normal(size=(n, 3)) * scale * 0.4
    L = np.array([0.3, 1.0, -0.2]) * scale + rng.normal(size=3)
    s = rng.uniform(-0.5, 1.5, size=(n, 1))
    c = p + (L - p) * s + rng.normal(size=(n, 3)) * rng.uniform(0.01, 3.0, size=(n, 1))
    lv = L - p
    m = np.sqrt((lv * lv).sum(axis=1) + 1.0)
    d, dw = lv / m[:, None], -1.0 / m
    Rstar = grazing_radius(p, d, dw, c)
    for delta in DELTAS:
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, dw, c, R) >= 0
        keep = FilterScene(c, R).tabulated_pass(p, L)
        assert not (hit & ~keep).any(), f"dropped {int((hit & ~keep).sum())} at delta={delta}"


def test_secondary_ray_filter_with_unnormalised_directions():
    rng = np.random.default_rng(7)
    n = 20000
    p = rng.normal(size=(n, 3)) * 40.0
    d = rng.normal(size=(n, 3)) * rng.uniform(0.01, 20.0, size=(n, 1))  # refraction directions are not unit
    c = p + d * rng.uniform(-3, 3, size=(n, 1)) + rng.normal(size=(n, 3)) * rng.uniform(0.01, 2.0, size=(n, 1))
    Rstar = grazing_radius(p, d, np.zeros(n), c)
    for delta in DELTAS:
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, np.zeros(n), c, R) >= 0
        keep = FilterScene(c, R).general_pass(p, d, np.zeros(n), pairwise=True)
        assert not (hit & ~keep).any()


def test_exact_model_agrees_with_the_oracle(rtk, oracle):
    rng = np.random.default_rng(3)
    for _ in range(200):
        c, R = rng.normal(size=3) * 5, rng.uniform(0.2, 2.0)
        p, d = rng.normal(size=3) * 5, rng.normal(size=3)
        dw = rng.choice([0.0, -0.1, 0.3])
        node = rtk.Sphere().with_transform(rtk.Matrix4x4.translate(*c) * rtk.Matrix4x4.scale(R, R, R))
        roots = oracle.object_intersect(node, (*p, 1.0), (*d, dw))
        disc = exact_disc(p[None], d[None], np.array([dw]), c[None], np.array([R]))[0]
        assert (len(roots) == 2) == (disc >= 0) or abs(disc) < 1e-9


def test_filter_stays_selective_on_a_c4_like_scene(rtk):
    """Regression guard for speed: on the 10k-sphere layout the filter keeps only a small
    multiple of the true hits (every kept pair costs an f64 re-evaluation)."""
    scene, camera = rtk.synthetic.config("C4", 64, 36)
    spheres = [n for n in scene.nodes if n.kind == rtk.abi.RTK_SPHERE]
    c = np.array([[n.transform.e[3], n.transform.e[7], n.transform.e[11]] for n in spheres])
    R = np.array([n.transform.e[0] for n in spheres])
    fs = FilterScene(c, R)
    rng = np.random.default_rng(1)
    # camera rays through random pixels
    inv = np.linalg.inv(np.array(camera.transform.e).reshape(4, 4))
    origin = inv[:3, 3]
    xs, ys = rng.uniform(0, camera.width, 256), rng.uniform(camera.height * 0.3, camera.height, 256)
    px = np.stack([camera.half_width - xs * camera.pixel_size, camera.half_height - ys * camera.pixel_size,
                   -np.ones(256), np.ones(256)], axis=1)
    d = _unit((inv @ px.T).T[:, :3] - origin)
    keep = fs.primary_pass(origin, d)
    hit = np.stack([exact_disc(np.broadcast_to(origin, (len(c), 3)), np.broadcast_to(di, (len(c), 3)), np.zeros(len(c)), c, R) >= 0
                    for di in d])
    assert not (hit & ~keep).any()
    assert keep.sum() <= 2.0 * hit.sum() + 64, (int(keep.sum()), int(hit.sum()))
    # shadow rays from points on the floor towards the first light
    L = np.array(scene.lights[0].position[:3])
    p = np.stack([rng.uniform(-60, 60, 256), np.full(256, 1e-4), rng.uniform(0, 120, 256)], axis=1)
    keep = fs.general_pass(p, np.broadcast_to(L, p.shape), np.full(256, -1.0))
    lv = L - p
    m = np.sqrt((lv * lv).sum(axis=1) + 1.0)
    hit = np.stack([exact_disc(np.broadcast_to(p[i], (len(c), 3)), np.broadcast_to(lv[i] / m[i], (len(c), 3)),
                               np.full(len(c), -1.0 / m[i]), c, R) >= 0 for i in range(256)])
    assert not (hit & ~keep).any()
    assert keep.sum() <= 2.0 * hit.sum() + 64, (int(keep.sum()), int(hit.sum()))
    # the tabulated form trades some selectivity for speed; it must stay within a small factor
    keep_t = np.stack([fs_rows(fs, p[i], L, len(c)) for i in range(64)])
    assert not (hit[:64] & ~keep_t).any()
    assert keep_t.sum() <= 3.0 * hit[:64].sum() + 64, (int(keep_t.sum()), int(hit[:64].sum()))


def fs_rows(fs, point, light, n):
    """Tabulated filter of one shading point against all n spheres."""
    return fs.tabulated_pass(np.broadcast_to(point, (n, 3)), light)


# ---- wavefront-level tests -------------------------------------------------------------------
from filter_model import (camera_records, cluster_balls, cone_reaches, lines_reach, make_ball, make_cone,
                          sphere_balls)


@pytest.mark.parametrize("scale", [3.0, 60.0, 1500.0])
def test_cone_test_keeps_every_sphere_some_ray_of_the_patch_grazes(scale):
    """64 camera rays of an 8x8 patch; 64 spheres, sphere i only just touched by ray i.  Neither
    the sphere-level nor the cluster-level cone test may drop one."""
    rng = np.random.default_rng(int(scale) + 11)
    for trial in range(60):
        origin = rng.normal(size=3) * scale * 0.3
        axis = _unit(rng.normal(size=3))
        e1 = _unit(np.cross(axis, rng.normal(size=3)))
        e2 = np.cross(axis, e1)
        ps = 10.0 ** rng.uniform(-4.5, -1.5)  # pixel pitch on the image plane
        off = rng.uniform(-0.8, 0.8, size=2)
        gx, gy = np.meshgrid(np.arange(8) - 3.5, np.arange(8) - 3.5)
        d = _unit(axis + e1 * (off[0] + gx.reshape(-1, 1) * ps) + e2 * (off[1] + gy.reshape(-1, 1) * ps))
        n = 64
        dist = scale * rng.uniform(0.02, 1.0, size=(n, 1))
        perp = _unit(np.cross(d, rng.normal(size=(n, 3))))
        c = origin + d * dist + perp * rng.uniform(0.01, 2.0, size=(n, 1))
        p = np.broadcast_to(origin, (n, 3))
        Rstar = grazing_radius(p, d, np.zeros(n), c)
        cone = make_cone(d.astype(np.float32))
        for delta in (1e-13, 1e-8, 1e-3):
            R = Rstar * (1.0 + delta)
            hit = exact_disc(p, d, np.zeros(n), c, R) >= 0
            fs = FilterScene(c, R)
            spheres = camera_records(c - origin, R)
            assert not (hit & ~cone_reaches(cone, spheres)).any(), (trial, delta)
            assert not (hit & ~cone_reaches(make_cone(d.astype(np.float32), inv_ulps=1), spheres)).any(), (trial, delta)
            balls = sphere_balls(fs)
            cl = cluster_balls(balls[:, :3], balls[:, 3])
            Lc = cl[:, :3].astype(np.float64) + fs.center - origin
            assert cone_reaches(cone, camera_records(Lc, cl[:, 3].astype(np.float64))).all() or not hit.any()


@pytest.mark.parametrize("scale", [3.0, 60.0, 1500.0])
def test_line_bundle_test_keeps_every_sphere_some_shadow_ray_of_the_patch_grazes(scale):
    rng = np.random.default_rng(int(scale) + 23)
    for trial in range(60):
        n = 64
        centre = rng.normal(size=3) * scale * 0.4
        pts = centre + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 0.3)
        nl = int(rng.integers(1, 5))
        lights = np.array([0.3, 1.0, -0.2]) * scale * rng.uniform(0.3, 3.0) + rng.normal(size=(nl, 3)) * scale * 0.3
        k = rng.integers(0, nl, size=n)
        L = lights[k]
        lv = L - pts
        m = np.sqrt((lv * lv).sum(axis=1) + 1.0)
        d, dw = lv / m[:, None], -1.0 / m
        s = rng.uniform(-0.5, 1.5, size=(n, 1))
        c = pts + lv * s + rng.normal(size=(n, 3)) * rng.uniform(0.01, 3.0, size=(n, 1))
        Rstar = grazing_radius(pts, d, dw, c)
        for delta in (1e-13, 1e-8, 1e-3):
            R = Rstar * (1.0 + delta)
            hit = exact_disc(pts, d, dw, c, R) >= 0
            fs = FilterScene(c, R)
            ball = make_ball((pts - fs.center).astype(np.float32))
            balls = sphere_balls(fs)
            assert not (hit & ~lines_reach(fs, ball, lights, balls, cluster=False)).any(), (trial, delta)
            cl = cluster_balls(balls[:, :3], balls[:, 3])
            assert lines_reach(fs, ball, lights, cl, cluster=True).all() or not hit.any()


def test_bundle_tests_stay_selective_on_a_c4_like_scene(rtk):
    """Regression guard for speed: an 8x8 patch looking at the 10k-sphere field keeps few clusters."""
    scene, camera = rtk.synthetic.config("C4", 3840, 2160)
    spheres = [n for n in scene.nodes if n.kind == rtk.abi.RTK_SPHERE]
    c = np.array([[n.transform.e[3], n.transform.e[7], n.transform.e[11]] for n in spheres])
    R = np.array([n.transform.e[0] for n in spheres])
    inv = np.linalg.inv(np.array(camera.transform.e).reshape(4, 4))
    origin = inv[:3, 3]
    gx, gy = np.meshgrid(np.arange(8), np.arange(8))
    kept = []
    for x0, y0 in ((1900, 1500), (600, 1900), (3000, 1200)):
        xs, ys = (x0 + gx).reshape(-1) + 0.5, (y0 + gy).reshape(-1) + 0.5
        px = np.stack([camera.half_width - xs * camera.pixel_size, camera.half_height - ys * camera.pixel_size,
                       -np.ones(64), np.ones(64)], axis=1)
        d = _unit((inv @ px.T).T[:, :3] - origin)
        keep = cone_reaches(make_cone(d.astype(np.float32)), camera_records(c - origin, R))
        hit = np.stack([exact_disc(np.broadcast_to(origin, (len(c), 3)), np.broadcast_to(di, (len(c), 3)),
                                   np.zeros(len(c)), c, R) >= 0 for di in d]).any(axis=0)
        assert not (hit & ~keep).any()
        kept.append(int(keep.sum()))
    assert max(kept) <= 64, kept


# ---- per-ray cluster tests (one ray per lane: secondary rays, shadow rays of any light) ------
from filter_model import camera_ray_reaches_cluster, ray_reaches_cluster


def _random_clusters_around(rng, c, R, per_cluster=8):
    """Clusters that each contain one of the given spheres plus a few random neighbours."""
    n = len(c)
    others_c = c[:, None, :] + rng.normal(size=(n, per_cluster - 1, 3)) * rng.uniform(0.1, 5.0, size=(n, 1, 1))
    others_R = rng.uniform(0.05, 2.0, size=(n, per_cluster - 1))
    return np.concatenate([c[:, None, :], others_c], axis=1), np.concatenate([R[:, None], others_R], axis=1)


@pytest.mark.parametrize("light_w", [0.0, 1.0, 0.5])
@pytest.mark.parametrize("scale", [2.0, 50.0, 400.0])
def test_per_ray_cluster_test_keeps_every_grazing_hit(scale, light_w):
    rng = np.random.default_rng(int(scale * 7 + light_w * 5))
    n = 1500
    p = rng.normal(size=(n, 3)) * scale * 0.4
    L = np.array([0.3, 1.0, -0.2]) * scale + rng.normal(size=3)
    s = rng.uniform(-0.5, 1.5, size=(n, 1))
    c = p + (L - p) * s + rng.normal(size=(n, 3)) * rng.uniform(0.01, 3.0, size=(n, 1))
    lv = L - p
    lw = light_w - 1.0
    m = np.sqrt((lv * lv).sum(axis=1) + lw * lw)
    d, dw = lv / m[:, None], lw / m
    Rstar = grazing_radius(p, d, dw, c)
    G = L - light_w * p
    for delta in (1e-13, 1e-8, 1e-3):
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, dw, c, R) >= 0
        cc, RR = _random_clusters_around(rng, c, R)
        fs = FilterScene(cc.reshape(-1, 3), RR.reshape(-1))
        balls = sphere_balls(fs)
        clusters = cluster_balls(balls[:, :3], balls[:, 3], bundle=cc.shape[1])  # cluster i holds sphere i first
        reach = ray_reaches_cluster(fs, p, G, np.full(n, lw), clusters)
        own = reach[np.arange(n), np.arange(n)]
        assert not (hit & ~own).any(), f"dropped {int((hit & ~own).sum())} at delta={delta}"
    # and the test is not vacuous: a plain ray (w = 1) skips most foreign clusters.  (With w < 1 the
    # adversarial radii here, up to half the distance to the light, make the growing balls reach far.)
    assert light_w < 1.0 or scale < 50 or reach.mean() < 0.5


@pytest.mark.parametrize("scale", [2.0, 50.0, 400.0, 3000.0])
def test_clusters_behind_the_origin_hold_no_positive_root(scale):
    """Closest-hit rays skip a cluster whose ball lies entirely behind the ray origin.  Adversarial
    spheres: centred behind the origin with the origin just inside their front surface, so their
    larger root is a tiny positive number; none of them may sit in a skipped cluster."""
    from filter_model import ray_cluster_behind

    rng = np.random.default_rng(int(scale) + 77)
    n = 1500
    p = rng.normal(size=(n, 3)) * scale * 0.4
    d = _unit(rng.normal(size=(n, 3))) * rng.uniform(0.2, 3.0, size=(n, 1))  # secondary rays are not unit
    back = rng.uniform(0.01, 4.0, size=(n, 1))
    c = p - _unit(d) * back + rng.normal(size=(n, 3)) * back * rng.uniform(0.0, 0.5, size=(n, 1))
    dist = np.sqrt(((p - c) ** 2).sum(axis=1))
    for delta in (1e-13, 1e-9, 1e-6, 1e-3):
        R = dist * (1.0 + delta)  # the origin is inside by delta: roots straddle zero
        oo, dd = (p - c) / R[:, None], d / R[:, None]
        a, b, cc = (dd * dd).sum(axis=1), 2.0 * (oo * dd).sum(axis=1), (oo * oo).sum(axis=1) - 1.0
        disc = b * b - 4.0 * a * cc
        t1 = (-b + np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a)
        positive = (disc >= 0) & (t1 > 0.0)
        assert positive.mean() > 0.9
        cc3, RR = _random_clusters_around(rng, c, R)
        fs = FilterScene(cc3.reshape(-1, 3), RR.reshape(-1))
        balls = sphere_balls(fs)
        clusters = cluster_balls(balls[:, :3], balls[:, 3], bundle=cc3.shape[1])  # cluster i holds sphere i first
        behind = ray_cluster_behind(fs, p, d, clusters)
        own = behind[np.arange(n), np.arange(n)]
        assert not (positive & own).any(), f"skipped {int((positive & own).sum())} spheres with a positive root at delta={delta}"
    # not vacuous: clusters well behind are skipped
    far = p - _unit(d) * (20.0 + scale)
    cc3, RR = _random_clusters_around(rng, far, np.full(n, 0.5))
    fs = FilterScene(cc3.reshape(-1, 3), RR.reshape(-1))
    balls = sphere_balls(fs)
    clusters = cluster_balls(balls[:, :3], balls[:, 3], bundle=cc3.shape[1])
    assert ray_cluster_behind(fs, p, d, clusters)[np.arange(n), np.arange(n)].mean() > 0.9


@pytest.mark.parametrize("scale", [3.0, 60.0, 1500.0])
def test_camera_ray_cluster_test_keeps_every_grazing_hit(scale):
    rng = np.random.default_rng(int(scale) + 31)
    n = 1500
    origin = rng.normal(size=3) * scale * 0.3
    d = _unit(rng.normal(size=(n, 3)))
    dist = scale * rng.uniform(0.02, 1.0, size=(n, 1))
    perp = _unit(np.cross(d, rng.normal(size=(n, 3))))
    c = origin + d * dist + perp * rng.uniform(0.01, 2.0, size=(n, 1))
    p = np.broadcast_to(origin, (n, 3))
    Rstar = grazing_radius(p, d, np.zeros(n), c)
    for delta in (1e-13, 1e-8, 1e-3):
        R = Rstar * (1.0 + delta)
        hit = exact_disc(p, d, np.zeros(n), c, R) >= 0
        cc, RR = _random_clusters_around(rng, c, R)
        fs = FilterScene(cc.reshape(-1, 3), RR.reshape(-1))
        balls = sphere_balls(fs)
        cl = cluster_balls(balls[:, :3], balls[:, 3], bundle=cc.shape[1])
        rec = camera_records(cl[:, :3].astype(np.float64) + fs.center - origin, cl[:, 3].astype(np.float64))
        reach = camera_ray_reaches_cluster(d, rec)
        own = reach[np.arange(n), np.arange(n)]
        assert not (hit & ~own).any(), f"dropped {int((hit & ~own).sum())} at delta={delta}"
    assert scale < 50 or reach.mean() < 0.5


# ---- cone of shadow rays towards a light built from a point (w = 1) -----------------------------
from filter_model import light_cone_reaches


@pytest.mark.parametrize("scale", [3.0, 60.0, 1500.0])
def test_light_cone_keeps_every_sphere_some_shadow_ray_of_the_patch_grazes(scale):
    rng = np.random.default_rng(int(scale) + 41)
    skipped_any = False
    for trial in range(60):
        n = 64
        centre = rng.normal(size=3) * scale * 0.4
        spread = 10.0 ** rng.uniform(-3, 0.3)
        pts = centre + rng.normal(size=(n, 3)) * spread
        # the light anywhere from inside the patch's ball to far away
        L = centre + _unit(rng.normal(size=3)) * spread * 10.0 ** rng.uniform(-0.5, 3.0)
        lv = L - pts
        mlen = np.sqrt((lv * lv).sum(axis=1))          # w = 1: the direction's w is 0
        d, dw = lv / mlen[:, None], np.zeros(n)
        s = rng.uniform(0.02, 0.98, size=(n, 1))        # occluders between point and light
        c = pts + lv * s + rng.normal(size=(n, 3)) * rng.uniform(0.01, 3.0, size=(n, 1))
        Rstar = grazing_radius(pts, d, dw, c)
        for delta in (1e-13, 1e-8, 1e-3):
            R = Rstar * (1.0 + delta)
            # the cone is one-sided: what counts is a root inside the segment, 0 < t < m
            # (src/scene.rs:217-222), not one on the line beyond the light
            oo, dd = (pts - c) / R[:, None], d / R[:, None]
            qa, qb, qc = (dd * dd).sum(axis=1), 2.0 * (oo * dd).sum(axis=1), (oo * oo).sum(axis=1) - 1.0
            disc = qb * qb - 4.0 * qa * qc
            root = np.sqrt(np.maximum(disc, 0.0))
            t0, t1 = (-qb - root) / (2.0 * qa), (-qb + root) / (2.0 * qa)
            hit = (disc >= 0) & (((t0 > 0) & (t0 < mlen)) | ((t1 > 0) & (t1 < mlen)))
            fs = FilterScene(c, R)
            ball = make_ball((pts - fs.center).astype(np.float32))
            balls = sphere_balls(fs)
            keep = light_cone_reaches(fs, ball, L, balls)
            assert not (hit & ~keep).any(), (trial, delta)
            cl = cluster_balls(balls[:, :3], balls[:, 3])
            assert light_cone_reaches(fs, ball, L, cl).all() or not hit.any()
            skipped_any |= bool((~keep).any())
            seen_hits = hit.any()
    # padding records (negative radius) never pass, not even when the light sits inside the ball
    fs = FilterScene(np.zeros((2, 3)) + [[0, 0, 0], [5, 0, 0]], np.ones(2))
    pad = np.array([[0, 0, 0, -1e30]], np.float32)
    for light in ([0.1, 0.0, 0.0], [50.0, 3.0, 1.0]):
        ball = make_ball(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32))
        assert not light_cone_reaches(fs, ball, light, pad).any()


# ---- spheres behind a shadow ray -------------------------------------------------------------------

def reference_quadratic(p, L, c, R):
    """b and c of the reference's object-space quadratic (src/scene/shapes/spheres.rs:18-35) for the
    shadow ray from p towards a w = 0 light at L (src/scene.rs:208-225) and a sphere translate(c) *
    scale(R), in f64, operation by operation; and whether the ray is blocked."""
    sigma = 1.0 / R
    tau = -(c * sigma[:, None])
    lv = L - p
    m = np.sqrt(lv[:, 0] * lv[:, 0] + lv[:, 1] * lv[:, 1] + lv[:, 2] * lv[:, 2] + 1.0)
    d = lv / m[:, None]
    dw = -1.0 / m
    oo = sigma[:, None] * p + tau
    dd = sigma[:, None] * d + tau * dw[:, None]
    a = dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2] + dw * dw
    b = 2.0 * (oo[:, 0] * dd[:, 0] + oo[:, 1] * dd[:, 1] + oo[:, 2] * dd[:, 2])
    cc = (oo[:, 0] * oo[:, 0] + oo[:, 1] * oo[:, 1] + oo[:, 2] * oo[:, 2]) - 1.0
    disc = b * b - 4.0 * a * cc
    with np.errstate(all="ignore"):
        s = np.sqrt(disc)
        t0, t1 = (-b - s) / (2.0 * a), (-b + s) / (2.0 * a)
        blocked = (disc >= 0.0) & (((t0 > 0.0) & (t0 < m)) | ((t1 > 0.0) & (t1 < m)))
    return b, cc, blocked


@pytest.mark.parametrize("offset", [0.0, 1.0e3, 1.0e6])
@pytest.mark.parametrize("scale", [2.0, 50.0, 400.0, 3000.0])
def test_behind_test_only_drops_spheres_with_no_positive_root(scale, offset):
    """Points a hair outside their sphere (the reference's 1e-4 offset, and far less), light directions
    on both sides of the tangent plane and a hair off it, scenes far from the origin: whatever the
    FP32 test drops has b >= 0 and c >= 0 in the reference's own f64 arithmetic, hence no positive root."""
    rng = np.random.default_rng(int(scale) + int(offset) % 97 + 11)
    n = 40000
    far = np.array([offset, -0.5 * offset, 0.25 * offset])
    c = rng.normal(size=(n, 3)) * scale * 0.4 + far
    R = rng.uniform(0.05, 3.0, size=n) * rng.choice([0.1, 1.0, 5.0], size=n)
    L = np.array([-0.6, 1.0, -0.5]) * (scale * 1.5 + 10.0) + far * rng.choice([0.0, 1.0])
    normal = _unit(rng.normal(size=(n, 3)))
    stand_off = rng.choice([1.0e-4, 1.0e-5, 1.0e-7, 1.0e-9, 1.0e-12, 0.0, -1.0e-9, -1.0e-4, 0.3], size=n)
    p = c + normal * (R + stand_off)[:, None]
    # a third of the points: turned so that u.(L - u) is (nearly) zero -- the terminator
    u = p - c
    t = _unit(np.cross(u, rng.normal(size=(n, 3))))
    w = L - u  # direction of the object-space ray up to sigma / m
    # rotate u within span(u, t) so that u'.(L - u') ~ 0: solve numerically by bisection on the angle
    lo, hi = np.zeros(n), np.full(n, np.pi)
    ru = np.linalg.norm(u, axis=1)
    def f(theta):
        uu = (np.cos(theta) * ru)[:, None] * _unit(u) + (np.sin(theta) * ru)[:, None] * t
        return (uu * (L - uu)).sum(axis=1), uu
    flo, _ = f(lo)
    fhi, _ = f(hi)
    usable = (flo > 0) != (fhi > 0)
    for _ in range(80):
        mid = 0.5 * (lo + hi)
        fm, _ = f(mid)
        same = (fm > 0) == (flo > 0)
        lo = np.where(same, mid, lo)
        hi = np.where(same, hi, mid)
    theta = 0.5 * (lo + hi) + rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3], size=n)
    _, uu = f(theta)
    pick = usable & (rng.uniform(size=n) < 0.33)
    p = np.where(pick[:, None], c + uu, p)
    fs = FilterScene(c, R)
    behind = fs.tabulated_behind(p, L)
    b, cc, blocked = reference_quadratic(p, L, c, R)
    assert behind.any()
    assert not (behind & blocked).any()
    assert (b[behind] >= 0.0).all() and (cc[behind] >= 0.0).all()


def test_behind_test_catches_the_lit_side_of_a_c4_like_scene():
    """10,000 spheres within +-60: of the surface points (offset 1e-4) that face the light by more than a
    few degrees, nearly all are recognised -- the exact test is left for the dark side and the terminator."""
    rng = np.random.default_rng(77)
    n = 10000
    c = rng.uniform(-60.0, 60.0, size=(n, 3))
    R = rng.uniform(0.3, 1.2, size=n)
    L = np.array([-100.0, 200.0, -100.0])
    normal = _unit(rng.normal(size=(n, 3)))
    p = c + normal * (R + 1.0e-4)[:, None]
    behind = FilterScene(c, R).tabulated_behind(p, L)
    # in the sphere's frame the ray leaves u = p - c along L - u (the direction's w = -1/m meets the
    # translation column of the inverse transform)
    u = p - c
    cos_l = (_unit(u) * _unit(L - u)).sum(axis=1)
    assert behind[cos_l > 0.01].mean() > 0.999
    assert not behind[cos_l < 0.0].any()
