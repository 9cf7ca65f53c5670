"""numpy float32 model of the device's FP32 candidate filter (test infrastructure).

Mirrors, operation for operation, the record construction in csrc/rtk_api.hip
(rtk_scene_upload), `prep_primary_kernel`, `make_filter_ray`, `filter_threshold`,
`sweep_primary` and `sweep_filter` in csrc/rtk_sweeps.h, and the tabulated shadow filter
(`PixelFilter`/`arm` and the hot loop of `shadow_uniform_kernel`) in csrc/rtk_opaque.hip.  A fused multiply-add is
modelled as an f64 multiply-add rounded once to f32 (the f64 product of two f32 values is
exact, so this differs from a true fma only in rare double-rounding cases).
"""
import numpy as np

F = np.float32
EPS32 = F(5.9604645e-8)
KAPPA = F(1.0) - F(24.0) * EPS32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(F)


def round_up(v):
    f = np.asarray(v, np.float64).astype(F)
    return np.where(f.astype(np.float64) < v, np.nextafter(f, F(np.inf)), f).astype(F)


class FilterScene:
    """Filter records for spheres given by centre [n,3] and bounding radius [n] (f64)."""

    def __init__(self, centres, radii):
        c = np.asarray(centres, np.float64).reshape(-1, 3)
        self.centres, self.radii = c, np.asarray(radii, np.float64).reshape(-1)
        self.center = 0.5 * (c.min(axis=0) + c.max(axis=0))
        rel = c - self.center
        self.cf = rel.astype(F)
        shift = np.sqrt(((rel - self.cf.astype(np.float64)) ** 2).sum(axis=1))
        R = self.radii * (1.0 + 1e-9) + shift * 1.0001
        self.r2 = round_up(R * R)
        self.r2_max = self.r2.max()
        self.r_max = round_up(R).max()
        self.c_max = round_up(np.sqrt((self.cf.astype(np.float64) ** 2).sum(axis=1)) * 1.0001).max()

    # ---- primary rays ------------------------------------------------------------
    def primary_records(self, origin):
        L = self.centres - np.asarray(origin, np.float64)
        LL = (L * L).sum(axis=1)
        eps = float(EPS32)
        R = self.radii * (1.0 + 1e-12) + 1.01 * eps * np.sqrt(LL)
        cc = LL * (1.0 - 12.0 * eps) - R * R
        ccf = cc.astype(F)
        ccf = np.where(ccf.astype(np.float64) > cc, np.nextafter(ccf, F(-np.inf)), ccf)
        ccf = np.nextafter(ccf, F(-np.inf)).astype(F)
        return L.astype(F), ccf

    def primary_pass(self, origin, directions, pairwise=False):
        """The filter keeps the pair: [rays, spheres] boolean, or [n] for ray i vs sphere i."""
        Lf, ccf = self.primary_records(origin)
        d = np.asarray(directions, np.float64).astype(F)
        if pairwise:
            Lx, Ly, Lz, cc, dx, dy, dz = Lf[:, 0], Lf[:, 1], Lf[:, 2], ccf, d[:, 0], d[:, 1], d[:, 2]
        else:
            Lx, Ly, Lz, cc = Lf[None, :, 0], Lf[None, :, 1], Lf[None, :, 2], ccf[None, :]
            dx, dy, dz = d[:, None, 0], d[:, None, 1], d[:, None, 2]
        t = (Lx * dx).astype(F)
        t = fma(Ly, dy, t)
        t = fma(Lz, dz, t)
        return fma(t, t, -cc) >= F(0.0)

    # ---- general / shadow rays -------------------------------------------------------
    def threshold(self, p, rho_max):
        pc = np.asarray(p, np.float64) - self.center
        pn = np.sqrt((pc * pc).sum(axis=-1))
        dp = pn * float(EPS32) * 1.0001
        B = float(self.r2_max) * np.asarray(rho_max, np.float64)
        r2s = 2.0 * float(self.r_max) * dp + dp * dp
        hmax = pn + float(self.c_max) + B
        slack = r2s * (1.0 + 2.0 * hmax * rho_max) + 64.0 * float(EPS32) * (B * B + float(self.r2_max))
        return (-(slack * 1.001).astype(F) - F(1e-30)).astype(F)

    def general_pass(self, p, G, omega, pairwise=False):
        """p [rays,3] origins, G [rays,3] line directions, omega [rays]; -> [rays, spheres]
        (or [n] for ray i vs sphere i)."""
        p = np.asarray(p, np.float64)
        G = np.asarray(G, np.float64)
        omega = np.broadcast_to(np.asarray(omega, np.float64), p.shape[:1])
        inv = 1.0 / np.sqrt((G * G).sum(axis=1))
        g = (G * inv[:, None]).astype(F)
        rho = (omega * inv).astype(F)
        rho_max = (np.abs(rho) * F(1.0001)).astype(F)
        thr = self.threshold(p, rho_max.astype(np.float64))
        pc = (p - self.center).astype(F)
        if pairwise:
            ex = lambda a: a
            cfx, cfy, cfz, r2 = self.cf[:, 0], self.cf[:, 1], self.cf[:, 2], self.r2
        else:
            ex = lambda a: a[:, None]
            cfx, cfy, cfz, r2 = self.cf[None, :, 0], self.cf[None, :, 1], self.cf[None, :, 2], self.r2[None, :]
        ux = (ex(pc[:, 0]) - cfx).astype(F)
        uy = (ex(pc[:, 1]) - cfy).astype(F)
        uz = (ex(pc[:, 2]) - cfz).astype(F)
        uu = fma(uz, uz, fma(uy, uy, (ux * ux).astype(F)))
        cf = fma(uu, KAPPA, -r2)
        h = fma(uz, ex(g[:, 2]), fma(uy, ex(g[:, 1]), fma(ux, ex(g[:, 0]), (r2 * ex(rho)).astype(F))))
        return fma(h, h, -cf) >= ex(thr)

    # ---- tabulated shadow filter (lights with w = 0) ---------------------------------
    def tabulated_pass(self, p, light_xyz, pairwise=True):
        """Shadow ray from p [rays,3] towards ONE w = 0 light at light_xyz; ray i vs sphere i."""
        p = np.asarray(p, np.float64)
        Lv = np.asarray(light_xyz, np.float64)
        inv = 1.0 / np.sqrt((Lv * Lv).sum())
        gd = Lv * inv
        rho_d = -inv
        lf = np.array([gd[0], gd[1], gd[2], rho_d]).astype(F)
        # records built at scene upload
        cfd = self.cf.astype(np.float64)
        cc = float(KAPPA) * (cfd * cfd).sum(axis=1) - self.r2.astype(np.float64)
        ccf = cc.astype(F)
        ccf = np.where(ccf.astype(np.float64) > cc, np.nextafter(ccf, F(-np.inf)), ccf)
        ccf = np.nextafter(ccf, F(-np.inf)).astype(F)
        K = (cfd @ gd - self.r2.astype(np.float64) * rho_d).astype(F)
        # per-pixel set-up (arm)
        pc = p - self.center
        x, y, z = pc[:, 0].astype(F), pc[:, 1].astype(F), pc[:, 2].astype(F)
        P = fma(z, lf[2], fma(y, lf[1], (x * lf[0]).astype(F)))
        rho_max = float(np.abs(lf[3]) * F(1.0001))
        pn2 = (pc * pc).sum(axis=1)
        pn = np.sqrt(pn2)
        B = float(self.r2_max) * rho_max
        dh = float(EPS32) * (6.0 * pn + 2.0 * float(self.c_max) + 2.0 * B) * 1.01
        hmax = pn + float(self.c_max) + B
        cm2 = float(self.c_max) ** 2
        extra = 2.0 * hmax * dh + dh * dh + 6.0 * float(EPS32) * (pn2 + cm2 + float(self.r2_max)) * 1.01
        thr = self.threshold(p, np.full(len(p), rho_max)).astype(np.float64) - extra * 1.001
        pp = x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2 + z.astype(np.float64) ** 2
        t = float(KAPPA) * pp + thr
        tf = t.astype(F)
        tf = np.where(tf.astype(np.float64) > t, np.nextafter(tf, F(-np.inf)), tf)
        thrp = np.nextafter(tf, F(-np.inf)).astype(F)  # the device lowers tf by two ulps or more: never above this
        # hot loop
        two_kappa = F(2.0) * KAPPA
        assert pairwise
        pcd = fma(z, self.cf[:, 2], fma(y, self.cf[:, 1], (x * self.cf[:, 0]).astype(F)))
        e = fma(pcd, two_kappa, -ccf)
        h = (P - K).astype(F)
        return fma(h, h, e) >= thrp


    # ---- spheres behind a shadow ray (BehindMargins and the hot loop of shadow_phase_uniform) -------
    def tabulated_behind(self, p, light_xyz):
        """The device drops ray i vs sphere i as 'behind': shadow ray from p [rays,3] towards ONE w = 0
        light.  Margins per pixel (the device evaluates them once per wavefront, at the largest |pc| of its
        pixels; every term grows with |pc|, so its margins are never smaller)."""
        p = np.asarray(p, np.float64)
        Lv = np.asarray(light_xyz, np.float64)
        inv = 1.0 / np.sqrt((Lv * Lv).sum())
        gd = Lv * inv
        rho_d = -inv
        lf = np.array([gd[0], gd[1], gd[2], rho_d]).astype(F)
        cfd = self.cf.astype(np.float64)
        K = (cfd @ gd - self.r2.astype(np.float64) * rho_d).astype(F)
        ball_r = round_up(np.sqrt(self.r2.astype(np.float64)) * (1.0 + 1e-7))   # sph_ball.w
        pc = p - self.center
        x, y, z = pc[:, 0].astype(F), pc[:, 1].astype(F), pc[:, 2].astype(F)
        P = fma(z, lf[2], fma(y, lf[1], (x * lf[0]).astype(F)))
        lrho = np.abs(lf[3]) * F(1.0001)
        rho_max = float(lrho)
        pn = np.sqrt((pc * pc).sum(axis=1))
        B = float(self.r2_max) * rho_max
        dh = float(EPS32) * (6.0 * pn + 2.0 * float(self.c_max) + 2.0 * B) * 1.01
        hmax = pn + float(self.c_max) + B
        e32 = float(EPS32)
        reach = pn * (1.0 + e32) + float(self.c_max)
        absolute = reach + np.abs(self.center).sum()
        dp = pn * e32 * 1.0001
        dd = dp + e32 * float(self.c_max) * 1.01

        def above(v):
            return np.nextafter(round_up(v), F(np.inf)).astype(F)
        q1 = above(dp * 1.01 + 1.0e-12 * (absolute + reach))
        q3 = above(dh * 1.01 + dd * (1.0 + rho_max * (2.0 * reach + dd)) * 1.01 +
                   1.01 * e32 * (hmax + rho_max * (2.0 * reach * reach + float(self.r2_max))) +
                   1.0e-12 * reach * (1.0 + 2.0 * rho_max * absolute))
        # hot loop
        ux, uy, uz = (x - self.cf[:, 0]).astype(F), (y - self.cf[:, 1]).astype(F), (z - self.cf[:, 2]).astype(F)
        uu = fma(uz, uz, fma(uy, uy, (ux * ux).astype(F)))
        dist = np.nextafter(np.sqrt(uu.astype(np.float64)).astype(F), F(np.inf))  # v_sqrt_f32 is good to 1 ulp: take the worst case
        one_lo = F(1.0) - F(8.0) * EPS32
        outside = fma(dist, one_lo, -ball_r) >= q1
        rr = ((ball_r * ball_r).astype(F) * (F(1.0) - F(1.0e-6))).astype(F)
        gap = (fma(uu, F(1.0) + F(8.0) * EPS32, -rr) * (F(1.0) + F(4.0) * EPS32)).astype(F)
        h = (P - K).astype(F)
        return outside & (fma(-lrho, gap, h) >= q3)

# ---- wavefront-level tests (one sphere or cluster per lane against a bundle of 64 rays) -------

def cluster_balls(cf, ball_r, bundle=64):
    """Cluster records of rtk_scene_upload: mean centre and reach over aligned runs of `bundle`."""
    out = []
    for lo in range(0, len(cf), bundle):
        c, r = cf[lo:lo + bundle].astype(np.float64), ball_r[lo:lo + bundle].astype(np.float64)
        mf = c.mean(axis=0).astype(F)
        reach = (np.sqrt(((c - mf.astype(np.float64)) ** 2).sum(axis=1)) + r).max()
        out.append([mf[0], mf[1], mf[2], round_up(reach * (1.0 + 1e-7))])
    return np.array(out, dtype=F)


def sphere_balls(fs):
    """sph_ball records: recentred centre and the inflated radius, not squared."""
    return np.concatenate([fs.cf, round_up(np.sqrt(fs.r2.astype(np.float64)) * (1.0 + 1e-7))[:, None]], axis=1).astype(F)


def camera_records(L, radius):
    """prim_ball / prim_cluster records of prep_primary_kernel for balls at L (f64, relative to the
    camera origin) with the given radius."""
    Rs = radius * (1.0 + 1e-6) + 16.0 * float(EPS32) * np.sqrt((L * L).sum(axis=1))
    return np.concatenate([L.astype(F), round_up(Rs)[:, None]], axis=1).astype(F)


def ulps(v, k):
    """v moved by k FP32 steps (the hardware's one-instruction rsq, sqrt and rcp are good to an ulp)."""
    v = np.asarray(v, F)
    for _ in range(abs(k)):
        v = np.nextafter(v, F(np.inf) if k > 0 else F(-np.inf)).astype(F)
    return v


def make_cone(d, inv_ulps=-1):
    """make_cone of csrc/rtk_opaque.hip for the FP32 directions d [64, 3] of one 8x8 patch (row-major).  The
    axis lies between the directions of the two pixels either side of the patch's centre; the hardware's
    square roots and reciprocals are taken an ulp towards the narrower cone (inv_ulps: the axis length either way)."""
    d = d.astype(F)
    s = (d[27] + d[36]).astype(F)
    inv = ulps(F(1.0) / np.sqrt(F(s[0] * s[0] + s[1] * s[1] + s[2] * s[2])), inv_ulps)
    a = (s * inv).astype(F)
    along = (d[:, 0] * a[0] + d[:, 1] * a[1] + d[:, 2] * a[2]).astype(F)
    q = (d - along[:, None] * a[None, :]).astype(F)
    across = ulps(np.sqrt((q * q).sum(axis=1, dtype=F)).astype(F), -1)
    rcp = ulps((F(1.0) / np.where(along > 0, along, F(1))).astype(F), -1)
    t = np.where(along > 0, (across * rcp).astype(F), F(3.0e38)).astype(F)
    tan_t = F(t.max() * F(1.0 + 1.0e-5) + F(2.0e-6))
    sec_t = F(ulps(np.sqrt(F(1.0) + tan_t * tan_t), -1) * F(1.0 + 1.0e-6))
    return a, tan_t, sec_t


def cone_reaches(cone, ball):
    a, tan_t, sec_t = cone
    s = fma(ball[:, 2], a[2], fma(ball[:, 1], a[1], (ball[:, 0] * a[0]).astype(F)))
    qx, qy, qz = fma(-s, a[0], ball[:, 0]), fma(-s, a[1], ball[:, 1]), fma(-s, a[2], ball[:, 2])
    d2 = fma(qz, qz, fma(qy, qy, (qx * qx).astype(F)))
    rr = fma(s, tan_t, (ball[:, 3] * sec_t).astype(F))
    return (rr >= 0) & (fma(rr, rr, -d2) >= 0)


def make_ball(pts):
    """make_ball: mean and radius of the FP32 recentred shading points [n, 3] of one patch."""
    pts = pts.astype(F)
    n = F(len(pts))
    inv = F(1.0) / n
    m = (pts.sum(axis=0, dtype=F) * inv).astype(F)
    e = (pts - m).astype(F)
    far = ulps(np.sqrt((e * e).sum(axis=1, dtype=F)).astype(F), -1)  # the hardware's square root, an ulp short
    return m, F(far.max() * F(1.0 + 1.0e-5))


def lines_reach(fs, ball, lights_xyz, rec, cluster):
    """lines_reach of shadow_uniform_kernel: ball = make_ball(...), rec [n, 4] sphere or cluster
    records; True where some light's line bundle may touch the record's ball."""
    m, radius = ball
    Lv = np.asarray(lights_xyz, np.float64).reshape(-1, 3)
    inv = 1.0 / np.sqrt((Lv * Lv).sum(axis=1))
    g = (Lv * inv[:, None]).astype(F)
    rho = (-inv).astype(F)
    lrho = (np.abs(rho) * F(1.0001)).astype(F)
    grow = F(0)
    for k in range(len(Lv)):
        grow = max(grow, F(F(0.5) * fs.r_max * fs.r2_max * rho[k] * rho[k]))
    pad = F(radius + grow * F(1.01) +
            F(16.0) * EPS32 * (abs(m[0]) + abs(m[1]) + abs(m[2]) + radius + fs.c_max))
    t_extra = rec[:, 3] if cluster else F(0)
    r_grow = fs.r_max if cluster else rec[:, 3]
    ux, uy, uz = (m[0] - rec[:, 0]).astype(F), (m[1] - rec[:, 1]).astype(F), (m[2] - rec[:, 2]).astype(F)
    e4 = F(F(1.0) + F(4.0) * EPS32)
    fold = F(F(F(F(1.0) + F(8.0) * EPS32) * e4) * e4)
    uu = fma(uz, uz, fma(uy, uy, (ux * ux).astype(F)))
    reach = ((rec[:, 3] + pad).astype(F) * e4).astype(F)
    tb = (radius + t_extra).astype(F) if cluster else F(radius)
    best = np.full(len(rec), F(-3.0e38), F)
    for k in range(len(Lv)):
        # squared distance from the line as |u|^2 - s^2 (one multiply-add per light); its first-order roundings
        # are covered by 32 eps |u|^2 below
        s = fma(uz, g[k, 2], fma(uy, g[k, 1], (ux * g[k, 0]).astype(F)))
        t = np.maximum((tb - s).astype(F), F(0))
        rr = fma(t, (r_grow * F(lrho[k] * fold)).astype(F), reach)
        best = np.maximum(best, fma(rr, rr, fma(s, s, -uu)))
    return (reach >= 0) & (fma(uu, F(32.0) * EPS32, best) >= 0)


# ---- per-ray cluster tests of the one-ray-per-lane sweeps (csrc/rtk_sweeps.h) ------------------

def ray_reaches_cluster(fs, p, G, omega, clusters):
    """Cluster test of sweep_filter: ray i (origin p[i], line direction G[i], growth omega[i]/|G[i]|)
    against every cluster record [m, 4]; -> [rays, clusters]."""
    p = np.asarray(p, np.float64)
    G = np.asarray(G, np.float64)
    omega = np.broadcast_to(np.asarray(omega, np.float64), p.shape[:1])
    inv = 1.0 / np.sqrt((G * G).sum(axis=1))
    g = (G * inv[:, None]).astype(F)
    rho = (omega * inv).astype(F)
    lrho = (np.abs(rho) * F(1.0001)).astype(F)
    grow = (F(0.5) * fs.r_max * fs.r2_max * rho * rho).astype(F)
    pc = (p - fs.center).astype(F)
    pad = (grow * F(1.01) + F(16.0) * EPS32 * (np.abs(pc[:, 0]) + np.abs(pc[:, 1]) + np.abs(pc[:, 2]) + fs.c_max)).astype(F)
    cl = clusters[None, :, :]
    ux, uy, uz = (pc[:, None, 0] - cl[..., 0]).astype(F), (pc[:, None, 1] - cl[..., 1]).astype(F), (pc[:, None, 2] - cl[..., 2]).astype(F)
    room = (cl[..., 3] + pad[:, None]).astype(F)
    gx, gy, gz = g[:, None, 0], g[:, None, 1], g[:, None, 2]
    s = fma(uz, gz, fma(uy, gy, (ux * gx).astype(F)))
    qx, qy, qz = fma(-s, gx, ux), fma(-s, gy, uy), fma(-s, gz, uz)
    d2 = fma(qz, qz, fma(qy, qy, (qx * qx).astype(F)))
    t = (np.maximum((cl[..., 3] - s).astype(F), F(0)) * F(1.0 + 8.0 * float(EPS32))).astype(F)
    rr = (fma(t, (fs.r_max * lrho)[:, None].astype(F), room) * F(1.0 + 4.0 * float(EPS32))).astype(F)
    return (rr >= 0) & (fma(rr, rr, -d2) >= 0)


def ray_cluster_behind(fs, p, G, clusters):
    """`forward` part of sweep_filter's cluster test for plain rays (growth 0): True where the
    cluster ball lies entirely behind the origin, so none of its spheres has a positive root;
    -> [rays, clusters]."""
    p = np.asarray(p, np.float64)
    G = np.asarray(G, np.float64)
    inv = 1.0 / np.sqrt((G * G).sum(axis=1))
    g = (G * inv[:, None]).astype(F)
    pc = (p - fs.center).astype(F)
    pad = (F(16.0) * EPS32 * (np.abs(pc[:, 0]) + np.abs(pc[:, 1]) + np.abs(pc[:, 2]) + fs.c_max)).astype(F)
    cl = clusters[None, :, :]
    ux, uy, uz = (pc[:, None, 0] - cl[..., 0]).astype(F), (pc[:, None, 1] - cl[..., 1]).astype(F), (pc[:, None, 2] - cl[..., 2]).astype(F)
    room = (cl[..., 3] + pad[:, None]).astype(F)
    s = fma(uz, g[:, None, 2], fma(uy, g[:, None, 1], (ux * g[:, None, 0]).astype(F)))
    return s > room


def camera_ray_reaches_cluster(directions, records):
    """Cluster test of sweep_primary: unit FP32 directions [n, 3] against camera-side cluster
    records {L, Rs} [m, 4]; -> [n, m]."""
    d = np.asarray(directions, np.float64).astype(F)
    L = records[None, :, :]
    dx, dy, dz = d[:, None, 0], d[:, None, 1], d[:, None, 2]
    s = fma(L[..., 2], dz, fma(L[..., 1], dy, (L[..., 0] * dx).astype(F)))
    qx, qy, qz = fma(-s, dx, L[..., 0]), fma(-s, dy, L[..., 1]), fma(-s, dz, L[..., 2])
    d2 = fma(qz, qz, fma(qy, qy, (qx * qx).astype(F)))
    return (L[..., 3] * L[..., 3] * F(1.0 + 8.0 * float(EPS32))).astype(F) >= d2


# ---- cone of shadow rays towards a w = 1 light (shadow_phase_bundled in csrc/rtk_opaque.hip) ------

def light_cone_reaches(fs, ball, light_xyz, rec):
    """Patch ball = make_ball(...) of recentred points, one w = 1 light at light_xyz (world, f64),
    rec [n, 4] sphere or cluster records; True where a segment point -> light may touch the record."""
    m, radius = ball
    lf = (np.asarray(light_xyz, np.float64) - fs.center).astype(F)
    e = (m - lf).astype(F)
    D = np.sqrt(F(e[0] * e[0] + e[1] * e[1] + e[2] * e[2])).astype(F)
    apex_max = F(abs(lf[0]) + abs(lf[1]) + abs(lf[2]))
    rw = F(radius * F(1.0 + 1.0e-5) + F(8.0) * EPS32 * (apex_max + D))
    inv = F(1.0) / max(D, F(1.0e-30))
    a = (e * inv).astype(F)
    leg2 = F(D * D - rw * rw)
    slack = F(F(16.0) * EPS32 * (abs(m[0]) + abs(m[1]) + abs(m[2]) + radius + fs.c_max + apex_max))
    if not (leg2 > F(1.0e-6) * D * D):
        return rec[:, 3] >= 0
    tn = F(rw / np.sqrt(max(leg2, F(1.0e-30))) * F(1.0 + 1.0e-5) + F(2.0e-6))
    sec = F(np.sqrt(F(1.0) + tn * tn) * F(1.0 + 1.0e-6))
    lx, ly, lz = (rec[:, 0] - lf[0]).astype(F), (rec[:, 1] - lf[1]).astype(F), (rec[:, 2] - lf[2]).astype(F)
    s = fma(lz, a[2], fma(ly, a[1], (lx * a[0]).astype(F)))
    qx, qy, qz = fma(-s, a[0], lx), fma(-s, a[1], ly), fma(-s, a[2], lz)
    d2 = fma(qz, qz, fma(qy, qy, (qx * qx).astype(F)))
    rr = (fma(s, tn, ((rec[:, 3] + slack) * sec).astype(F)) * F(1.0 + 4.0 * float(EPS32))).astype(F)
    return (rr >= 0) & (fma(rr, rr, -d2) >= 0)
