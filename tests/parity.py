"""Shared parity checks: a backend under test vs the CPU oracle on the same scene description."""
import numpy as np

RGB_TOL = 1e-5  # BASELINE.json north_star: "pixels match the reference CPU render within 1e-5 per RGB channel"


def rgb_error(rgb, ref_rgb, label="", rel=False):
    """max |dRGB| over the finite channels; channels the reference makes NaN or infinite (0 * NaN blends at a NaN normal, specular
    overflow) must be the same NaN / the same infinity on the device.  rel: errors relative to max(1, |reference|) -- for ray sets
    with directions that are not unit vectors, where `reflect . eye` exceeds 1 and its power the range of a colour."""
    if not rgb.size:
        return 0.0
    odd_t, odd_r = ~np.isfinite(rgb), ~np.isfinite(ref_rgb)
    same = (odd_t == odd_r) & (~odd_r | (np.isnan(rgb) == np.isnan(ref_rgb))) & (~odd_r | np.isnan(ref_rgb) | (rgb == ref_rgb))
    assert same.all(), "%s: %d channels are NaN / infinite on one side only, first at %s: got %s want %s" % (
        label, int((~same).sum()), np.argwhere(~same)[:3].tolist(), rgb[~same][:3], ref_rgb[~same][:3])
    fin = ~odd_r
    if not fin.any():
        return 0.0
    d = np.abs(rgb[fin] - ref_rgb[fin])
    return float((d / np.maximum(1.0, np.abs(ref_rgb[fin]))).max() if rel else d.max())


def oracle_reference(orc, world, camera, fuel=5, pixel_indices=None, threads=0):
    """One oracle pass (rgb, primary hits, hit-tree digests), to be shared by several assert_parity calls on the same scene."""
    return orc.render_with_digest(orc.build_world(world), camera, fuel, pixel_indices, threads=threads)


def assert_parity(test_backend, orc, world, camera, fuel=5, pixel_indices=None, label="", digest=True, ref=None, rel=False):
    """Hit records must be bit-exact: the primary hit of every pixel (t as u64 bits, primitive sequence number, push index) and,
    through the hit-tree digest (include/rtc.h rtc_render_hit_digest), every closest hit of every pixel's ray tree — reflected and
    refracted rays at every depth; colours within RGB_TOL."""
    nw_t = test_backend.build_world(world)
    rgb, hits = test_backend.render(nw_t, camera, fuel, pixel_indices)
    ref_dig = None
    if ref is not None:
        ref_rgb, ref_hits, ref_dig = ref
    elif digest:
        ref_rgb, ref_hits, ref_dig = orc.render_with_digest(orc.build_world(world), camera, fuel, pixel_indices)
    else:
        ref_rgb, ref_hits = orc.render(orc.build_world(world), camera, fuel, pixel_indices)
    bad = (hits["prim"] != ref_hits["prim"]) | (hits["push_idx"] != ref_hits["push_idx"]) | (hits["t"].view(np.uint64) != ref_hits["t"].view(np.uint64))
    assert not bad.any(), "%s: %d/%d primary-hit records differ, first at %s: got %s want %s" % (
        label, int(bad.sum()), bad.size, np.flatnonzero(bad)[:3], hits[bad][:3], ref_hits[bad][:3])
    if ref_dig is not None:
        dig = test_backend.render_digest(nw_t, camera, fuel, pixel_indices)
        bad_d = dig != ref_dig
        assert not bad_d.any(), "%s: the hit-tree digests of %d/%d pixels differ (a closest hit somewhere below the primary one), first at %s" % (
            label, int(bad_d.sum()), bad_d.size, np.flatnonzero(bad_d)[:5])
    err = rgb_error(rgb, ref_rgb, label, rel)
    assert err <= RGB_TOL, "%s: max |dRGB| = %.3e > %.0e" % (label, err, RGB_TOL)
    return err


def assert_ray_parity(test_backend, orc, world, rays, fuel=5, label="", rel=False):
    nw_t, nw_o = test_backend.build_world(world), orc.build_world(world)
    rgb, hits = test_backend.color_at(nw_t, rays, fuel)
    ref_rgb, ref_hits = orc.color_at(nw_o, rays, fuel)
    bad = (hits["prim"] != ref_hits["prim"]) | (hits["push_idx"] != ref_hits["push_idx"]) | (hits["t"].view(np.uint64) != ref_hits["t"].view(np.uint64))
    assert not bad.any(), "%s: %d/%d hit records differ: got %s want %s" % (label, int(bad.sum()), bad.size, hits[bad][:3], ref_hits[bad][:3])
    err = rgb_error(rgb, ref_rgb, label, rel)
    assert err <= RGB_TOL, "%s: max |dRGB| = %.3e" % (label, err)
    return err


def _raises(backend, nw, rays, fuel):
    try:
        backend.color_at(nw, rays, fuel)
        return False
    except Exception as ex:  # noqa: BLE001 - both back ends raise RtwError; anything else is a finding too
        if "NaN" not in str(ex):
            raise
        return True


def assert_ray_parity_with_panics(test_backend, orc, world, rays, fuel=5, label="", max_panics=64, rel=False):
    """assert_ray_parity for ray sets that may hold rays on which the reference PANICS (a NaN t reaches its sort,
    src/intersection.rs:123-125): those rays are found with the oracle (bisection), the device must refuse each of them alone with
    RTC_ERR_NAN, and the rest of the set must match hit for hit.  Returns (max |dRGB|, number of panicking rays)."""
    nw_t, nw_o = test_backend.build_world(world), orc.build_world(world)
    panics = []

    def find(lo, hi):
        if len(panics) > max_panics or not _raises(orc, nw_o, rays[lo:hi], fuel):
            return
        if hi - lo == 1:
            panics.append(lo)
            return
        mid = (lo + hi) // 2
        find(lo, mid)
        find(mid, hi)

    find(0, len(rays))
    assert len(panics) <= max_panics, "%s: more than %d panicking rays" % (label, max_panics)
    for i in panics:
        assert _raises(test_backend, nw_t, rays[i:i + 1], fuel), "%s: the reference panics on ray %d %s, the device does not" % (label, i, rays[i])
    keep = np.ones(len(rays), dtype=bool)
    keep[panics] = False
    return assert_ray_parity(test_backend, orc, world, rays[keep], fuel, label=label, rel=rel), len(panics)
