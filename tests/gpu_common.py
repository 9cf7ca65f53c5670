"""Helpers shared by the GPU parity tests (comparison rules of the north star)."""
import numpy as np

REL_TOL = 1e-4          # per-pixel linear FP32 colour vs the f64 reference result
ABS_FLOOR = 1e-3        # relative error is taken against max(|reference|, ABS_FLOOR)
LSB_FRACTION = 0.9999   # share of pixels whose 8-bit channels must agree within 1 LSB


def compare_frames(gpu_rgb, gpu_rgba8, ref_rgb, ref_rgba8):
    """Returns (max relative error, pixels off by > REL_TOL, pixels off by > 1 LSB, pixels not bit-equal)."""
    with np.errstate(all="ignore"):
        rel = np.abs(gpu_rgb.astype(np.float64) - ref_rgb) / np.maximum(np.abs(ref_rgb), ABS_FLOOR)
        # The reference's Schlick term is unbounded, and a colour beyond the FP32 range narrows to the same
        # infinity on both sides: that is agreement, not an infinite error (a NaN stays a NaN and fails).
        same_inf = np.isinf(gpu_rgb) & (gpu_rgb == np.asarray(ref_rgb).astype(np.float32))
    rel = np.where(same_inf, 0.0, rel)
    rel = np.where(np.isnan(rel), np.inf, rel)
    d8 = np.abs(gpu_rgba8.astype(np.int32) - ref_rgba8.astype(np.int32)).max(axis=-1)
    return float(rel.max()), int((rel.max(axis=-1) > REL_TOL).sum()), int((d8 > 1).sum()), int((d8 > 0).sum())


def assert_parity(gpu_rgb, gpu_rgba8, ref_rgb, ref_rgba8, what=""):
    max_rel, bad_rel, bad_lsb, not_equal = compare_frames(gpu_rgb, gpu_rgba8, ref_rgb, ref_rgba8)
    n = ref_rgb.shape[0] * ref_rgb.shape[1]
    budget = int(n * (1.0 - LSB_FRACTION))
    assert bad_lsb <= budget, f"{what}: {bad_lsb} of {n} pixels differ by more than 1 LSB (budget {budget})"
    assert bad_rel <= budget, f"{what}: {bad_rel} of {n} pixels exceed {REL_TOL} relative (max {max_rel:.3e})"
    return max_rel, bad_rel, bad_lsb, not_equal
