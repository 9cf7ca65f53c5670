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
        thrp = np.nextafter(tf, F(-np.inf)).astype(F)
        # hot loop
        two_kappa = F(2.0) * KAPPA
        assert pairwise
        pcd = fma(z, self.cf[:, 2], fma(y, self.cf[:, 1], (x * self.cf[:, 0]).astype(F)))
        e = fma(pcd, two_kappa, -ccf)
        h = (P - K).astype(F)
        return fma(h, h, e) >= thrp
