"""rocprofv3 kernel names -> the names the library's launch profiler (csrc/runtime.hip, TG_LAUNCH) books them under."""
import re

_T = {"unsigned short": "bf16", "float": "f32"}


def tg_name(k):
    m = re.search(r"conv3x3_tile_kernel<([a-z ]+), ([a-z ]+), (\d+), (\d+), \d+(?:, (\d+))?>", k)
    if m:
        base = "conv3x3_tile<%s,%s,%s,%s" % (_T[m.group(1)], _T[m.group(2)], m.group(3), m.group(4))
        return base + (",pack%s>" % m.group(5) if m.group(5) and m.group(5) != "1" else ">")
    m = re.search(r"conv3x3_ws_kernel<(true|false), (true|false)(?:, (?:true|false))?(?:, (true|false))?>", k)
    if m:
        tags = [t for t, on in (("res", m.group(1)), ("aux", m.group(2)), ("frag", m.group(3))) if on == "true"]
        return "conv3x3_ws<%s>" % ",".join(tags)
    m = re.search(r"resblock_chain_kernel<(true|false)", k)                    # <HAS_AUX1, DIST, TR, SM>
    if m:
        return "resblock_chain<bwd>" if m.group(1) == "true" else "resblock_chain<fwd>"
    m = re.search(r"resblock_lat_kernel<(true|false), (true|false)", k)        # <HAS_AUX1, HAS_AUX2, FRAG, DIST>
    if m:
        return {("false", "false"): "resblock_lat<fwd>", ("true", "false"): "resblock_lat<bwd>",
                ("false", "true"): "resblock_lat<mask2>", ("true", "true"): "resblock_lat<bwd,mask2>"}[m.groups()]
    m = re.search(r"hr_fwd_lat_kernel<(true|false)(?:, (\d+), (\d+))?>", k)      # <FUSE, HF_TI, HF_TJ>
    if m:
        if m.group(1) == "false":
            return "hr_fwd_lat<deconv>"
        return "hr_fwd_lat<tail>" if m.group(2) in (None, "4") else "hr_fwd_lat<tail,%sx%s>" % (m.group(2), m.group(3))
    m = re.search(r"deconv_bwd_lat_kernel<(true|false)>", k)
    if m:
        return "deconv_bwd_lat<aux>" if m.group(1) == "true" else "deconv_bwd_lat<>"
    if "hr_bwd_lat_kernel" in k:
        return "hr_bwd_lat"
    m = re.search(r"conv3x3_dma_kernel<(true|false), (true|false)(?:, (\d+))?(?:, \d+)?>", k)
    if m:
        tags = [t for t, on in (("res", m.group(1)), ("aux", m.group(2))) if on == "true"]
        return "conv3x3_dma%s<%s>" % ({"2": "_pack2", "4": "_pack4"}.get(m.group(3), ""), ",".join(tags))
    m = re.search(r"conv3x3_wr_kernel<(true|false), (true|false), (\d+), (\d+), \d+>", k)     # <HAS_RES, HAS_AUX, TH, PK, KS>
    if m:
        tags = [t for t, on in (("res", m.group(1)), ("aux", m.group(2))) if on == "true"]
        base = "conv3x3_wr_pack2" if m.group(4) == "2" else ("conv3x3_wr" if m.group(3) == "16" else "conv3x3_wr8")
        return "%s<%s>" % (base, ",".join(tags))
    m = re.search(r"conv4x4s2_fwd_kernel<(true|false)>", k)
    if m:
        return "conv4x4s2_fwd<%s>" % ("res" if m.group(1) == "true" else "")
    m = re.search(r"conv4x4s2_bwd_kernel<(true|false), (true|false)>", k)
    if m:
        return "conv4x4s2_bwd<%s>" % ",".join(t for t, on in (("res", m.group(1)), ("aux", m.group(2))) if on == "true")
    if "resblock_thr_kernel" in k:
        return "resblock_thr"
    m = re.search(r"resblock_plane_kernel<\d+, \d+(?:, (true|false))?>", k)      # <D, L, PRE>
    if m:
        return "resblock_plane<in>" if m.group(1) == "true" else "resblock_plane"
    m = re.search(r"conv3x3_c8_kernel<(\d+)>", k)
    if m:
        return "conv3x3_c8<%s>" % m.group(1)
    m = re.search(r"conv_igemm_kernel<([a-z ]+), ([a-z ]+), (\d+), (\d+), (\d+), (\d+), (true|false)>", k)
    if m:
        return "conv_igemm<%s,%s,%s,%s,%s,%s>" % (_T[m.group(1)], _T[m.group(2)], m.group(3), m.group(4), m.group(5), m.group(6))
    if "hr_tail_kernel" in k:
        return "hr_tail"
    m = re.search(r"conv_wgrad_tr_kernel<(\d+)>", k)
    if m:
        return "conv_wgrad_tr" if m.group(1) == "64" else "conv_wgrad_tr_out"
    if "deconv3x3s2_ws_kernel" in k:
        return "deconv3x3s2_ws"
    if "conv_wgrad_row3_bf16_kernel" in k:
        return "conv_wgrad_row3_bf16"
    m = re.search(r"conv_wgrad_bf16_kernel<(\d+)>", k)
    if m:
        return "conv_wgrad_bf16<%s>" % m.group(1)
    m = re.search(r"conv_wgrad_kernel<([a-z ]+), ([a-z ]+)>", k)
    if m:
        return "conv_wgrad<%s,%s>" % (_T[m.group(1)], _T[m.group(2)])
    for base, tg in (("warp_s2d_fwd_scalar_kernel", "warp_s2d_fwd_scalar"), ("warp_s2d_fwd_kernel", "warp_s2d_fwd"),
                     ("warp_s2d_bwd_kernel", "warp_s2d_bwd"), 
                     ("bicubic_add_kernel", "bicubic_add"), ("upsample2_fwd_kernel", "upsample2_fwd")):
        m = re.search(base + r"<([a-z ]+)>", k)
        if m:
            return "%s<%s>" % (tg, _T[m.group(1)])
    m = re.search(r"bicubic_add_quad_kernel<([a-z ]+), (?:true|false)>", k)
    if m:
        return "bicubic_add_quad<%s>" % _T[m.group(1)]
    if "upsample2_fwd_x8_kernel" in k:
        return "upsample2_fwd_x8"
    return None
