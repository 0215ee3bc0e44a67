#!/usr/bin/env python3
"""Regenerates tests/golden/streams/*.npz.  Needs /root/reference (oracle/_ref is built from it): runs only in the build container.

Per case: a synthetic stream from the writer (openhevc_amd/synth/stream.c) is decoded TWICE by the reference's own decoder —
  * unmodified (oracle/_ref/libopenhevc_ref.so): the MD5 of every plane of every output picture is the expected result;
  * with this repository's recording hooks linked into its CTU loop (libopenhevc_hooked.so, oracle/ref_hooked_unit.c): the work
    list of every picture, exactly as the hooks recorded it inside the reference, is stored.
The fixtures let the CPU checker and the HIP engine be tested against REFERENCE OUTPUT of real bitstream decoding on machines where
the reference does not exist (the GPU box).  The streams themselves are regenerated on the fly (the writer is deterministic)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refdec        # noqa: E402
import streamgen     # noqa: E402
from openhevc_amd import frame as F    # noqa: E402

STREAM_CASES = [
    # name, width, height, seed, writer parameters
    ("ipb_8b", 416, 240, 7, dict(n_pictures=4, gop=2)),
    ("lowdelay_p_10b_ctb32_tools", 264, 200, 9, dict(n_pictures=3, gop=1, bit_depth=10, log2_ctb_size=5, pcm=1, transquant_bypass=1, transform_skip=1, cu_qp_delta=1)),
    ("b_ctb16_slices_wpp", 264, 200, 11, dict(n_pictures=3, gop=2, wpp=1, n_slices=3, log2_ctb_size=4, log2_max_tb_size=4)),
    ("b_tiles_slices_no_lf_across", 416, 240, 12, dict(n_pictures=3, gop=2, tile_cols=2, tile_rows=2, n_slices=3, log2_ctb_size=4, log2_max_tb_size=4,
                                                    lf_across_tiles=0, lf_across_slices=0)),
    ("p_weighted_cip_lists", 264, 200, 13, dict(n_pictures=3, gop=1, weighted_pred=1, scaling_list=1, constrained_intra_pred=1)),
    ("b_tmvp_3refs", 416, 240, 14, dict(n_pictures=4, gop=2, tmvp=1, cabac_init_present=1, deblocking_override=1, n_refs=3)),
    # 4:4:4 range extension with cross-component prediction (work lists carry the cross links the hooked decoder's shim records)
    ("b_444_ccp_8b", 264, 200, 15, dict(n_pictures=3, gop=2, chroma_format_idc=3, cross_component_pred=1, transform_skip=1, transquant_bypass=1)),
    ("i_444_ccp_10b_ctb16", 200, 136, 16, dict(n_pictures=2, gop=0, bit_depth=10, chroma_format_idc=3, cross_component_pred=1, log2_ctb_size=4, log2_max_tb_size=4)),
    # range-extension coding tools: transform skip up to 32x32 (rotation, own significance context), implicit / explicit residual DPCM on
    # skip and bypass blocks, persistent Rice adaptation, intra smoothing off
    ("b_rext_tskip_rdpcm_8b", 264, 200, 17, dict(n_pictures=3, gop=2, transform_skip=1, transquant_bypass=1, tskip_rotation=1, tskip_context=1, implicit_rdpcm=1,
                                                 explicit_rdpcm=1, log2_max_tskip_size=5, tskip_pct=50, bypass_pct=25)),
    ("i_444_rext_tools_10b", 200, 136, 18, dict(n_pictures=2, gop=0, bit_depth=10, chroma_format_idc=3, cross_component_pred=1, transform_skip=1, transquant_bypass=1,
                                                tskip_rotation=1, tskip_context=1, implicit_rdpcm=1, persistent_rice=1, intra_smoothing_disabled=1,
                                                log2_max_tskip_size=4, tskip_pct=40, bypass_pct=30)),
    # hierarchical B: decode order +4 +2 +1 +3, references from the future, sub-layer non-reference pictures, an IDR picture in the middle
    ("b_hier_tmvp_idr_10b", 264, 200, 19, dict(n_pictures=9, gop=3, bit_depth=10, tmvp=1, n_refs=3, idr_period=6)),
    # 4:2:2; and tiles with 16x16 CTBs, where the work lists carry the reference's filter-call order (OhFrame.sao_pending)
    ("b_422_tools_8b", 264, 200, 20, dict(n_pictures=3, gop=2, chroma_format_idc=2, transform_skip=1, transquant_bypass=1, tmvp=1)),
    ("p_422_ctb16_tiles_slices_10b", 200, 168, 21, dict(n_pictures=3, gop=1, chroma_format_idc=2, bit_depth=10, log2_ctb_size=4, log2_max_tb_size=4, n_slices=2,
                                                        sao_pct=90, tile_cols=2, tile_rows=2, lf_across_tiles=0)),
    ("i_420_ctb16_tiles3x2", 296, 168, 22, dict(n_pictures=2, gop=0, log2_ctb_size=4, log2_max_tb_size=4, sao_pct=90, tile_cols=3, tile_rows=2)),
]


def build(name, w, h, seed, kw):
    data, _ = streamgen.write_stream(w, h, seed, **kw)
    want = refdec.decode(data)
    out = {"stream_md5": np.frombuffer(hashlib.md5(data).digest(), dtype=np.uint8)}
    k = [0]

    def on_picture(f, cur, poc):
        for key, v in F.frame_to_arrays(f).items():
            out[f"pic{k[0]}_{key}"] = v
        k[0] += 1
    n = refdec.record_work_lists(data, on_picture)
    assert n == len(want), (n, len(want))
    out["n_pictures"] = np.array([n])
    # the digests in DECODE order, like the work lists (the reference hands its pictures out in output order)
    rank = streamgen.output_rank(n, kw.get("gop", 2), kw.get("idr_period", 0))
    out["md5"] = np.frombuffer(b"".join(b"".join(refdec.md5_of(want[rank[i]])) for i in range(n)), dtype=np.uint8).copy()
    return out


def main():
    only = set(sys.argv[1:])                                  # names: regenerate just these
    for name, w, h, seed, kw in STREAM_CASES:
        if only and name not in only:
            continue
        path = os.path.join(HERE, "streams", name + ".npz")
        np.savez_compressed(path, **build(name, w, h, seed, kw))
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
