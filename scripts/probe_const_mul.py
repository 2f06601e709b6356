"""Exploratory probe: constant multiplication of a batch of 256-bit field elements through the int8
matrix cores against the VALU product the NTT uses (jfhip_probe kinds 6 and 7, csrc/probe_mfma.hip)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import lib as L  # noqa: E402

L.load_library()
L.init(0)
res = {"v_mad_u64_u32_per_s": L.probe(0),
       "fp_mul_saturated_per_s": {"bn254_fr": L.probe(2, 0), "bls12_381_fr": L.probe(2, 2)},
       "f29_mul_per_s": {"bn254_fr": L.probe(3, 0), "bls12_381_fr": L.probe(3, 2)},
       "fp_mul_const29_per_s": {"bn254_fr": L.probe(6, 0), "bls12_381_fr": L.probe(6, 2)},
       "mfma_i8_const_mul_skeleton_per_s": L.probe(7)}
res["mfma_over_valu"] = res["mfma_i8_const_mul_skeleton_per_s"] / res["fp_mul_const29_per_s"]["bls12_381_fr"]
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "probe_const_mul.json"), "w"), indent=1)
