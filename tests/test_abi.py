"""CPU suite: the C-ABI library loads and exports every symbol include/jfhip.h declares, and the
host-side argument checks answer without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from jellyfish_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jfhip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load_library()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(L.SYMBOLS) == syms


def _c_arg_counts():
    text = open(os.path.join(ROOT, "include", "jfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(jfhip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_rust_sys_crate_declares_every_symbol_with_the_same_arity():
    """rust/jf-hip-sys cannot be compiled here (no Rust toolchain): keep it in step with the header textually --
    every function the header declares is an `pub fn` of the extern block with the same number of arguments, and the
    crate declares nothing the header does not."""
    src = open(os.path.join(ROOT, "rust", "jf-hip-sys", "src", "lib.rs")).read()
    src = re.sub(r"//.*", "", src)
    rust = {}
    for name, args in re.findall(r"pub fn (jfhip_[a-z0-9_]+)\s*\(([^)]*)\)", src, flags=re.S):
        args = args.strip().rstrip(",")
        rust[name] = 0 if not args else args.count(",") + 1
    c = _c_arg_counts()
    assert sorted(rust) == sorted(c)
    assert rust == c
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "rust", "jf-hip-sys", f))
    patch = open(os.path.join(ROOT, "rust", "patches", "jf-pcs-hip.patch")).read()
    for sym in ("jfhip_srs_load", "jfhip_msm", "jfhip_msm_batch", "jfhip_kzg_open", "jfhip_kzg_multi_point_open",
                "jfhip_last_error", "jfhip_srs_free", "jfhip_stream_create", "jfhip_stream_sync"):
        assert sym in patch and sym in c
    # what INTEGRATION.md says the Rust side holds: the GPUCommittable method set (reference mod.rs:659-965) for both
    # curves, the Drop hunk of srs.rs:40-44, the fingerprinted cache key
    for text in ("pub trait GPUCommittable<E: Pairing>", "impl<E: HipCurve> GPUCommittable<E> for UnivariateKzgPCS<E>",
                 "impl HipCurve for ark_bn254::Bn254", "impl HipCurve for ark_bls12_381::Bls12_381",
                 "fn gpu_commit(", "fn gpu_commit_with_loaded_prover_param(", "fn gpu_batch_commit(",
                 "fn gpu_batch_commit_with_loaded_prover_param(", "fn load_prover_param_to_gpu(", "fn load_poly_to_gpu",
                 "fn load_batch_poly_to_gpu", "fn commit_on_gpu(", "fn load_commitments_to_host(",
                 "impl<E: Pairing> Drop for UnivariateProverParam<E>", "fn fingerprint<E: Pairing>",
                 "hip::multi_point_open::<E>"):
        assert text in patch, text
    # the NTT call-site patches go through rust/jf-hip, whose device call is jfhip_ntt
    hip = open(os.path.join(ROOT, "rust", "jf-hip", "src", "lib.rs")).read()
    assert "jfhip_ntt(" in hip and "jfhip_ntt" in c
    for f in ("Cargo.toml",):
        assert os.path.exists(os.path.join(ROOT, "rust", "jf-hip", f))
    relation = open(os.path.join(ROOT, "rust", "patches", "jf-relation-hip.patch")).read()
    plonk = open(os.path.join(ROOT, "rust", "patches", "jf-plonk-hip.patch")).read()
    for text in ("jfhip_ntt", "jf_hip::ifft", "jf_hip::ifft_in_place"):
        assert text in relation, text
    # every iFFT site of the Arithmetization impl is swapped: 8 domain.ifft + 4 domain.ifft_in_place removed
    assert relation.count("-") and sum(1 for ln in relation.splitlines() if ln.startswith("-") and "domain.ifft(" in ln) == 8
    assert sum(1 for ln in relation.splitlines() if ln.startswith("-") and "domain.ifft_in_place(" in ln) == 4
    for text in ("jfhip_ntt", "jf_hip::coset_fft_batch", "jf_hip::coset_ifft", 'hip = ["dep:jf-hip", "jf-pcs/hip", "jf-relation/hip"]'):
        assert text in plonk, text


def test_no_device_is_reported_not_crashed():
    if L.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(L.JfHipError) as e:
        L.init(0)
    assert e.value.code == -3
    assert "HIP device" in e.value.msg


def test_argument_checks_without_gpu():
    lib = L.load_library()
    h = ctypes.c_void_p()
    assert lib.jfhip_srs_load(7, None, ctypes.c_size_t(4), ctypes.c_size_t(64), 0, ctypes.byref(h)) == -1
    assert lib.jfhip_msm(None, ctypes.c_size_t(0), None, ctypes.c_size_t(0), None, None, None) == -1
    assert "SRS" in L.last_error()
    assert lib.jfhip_msm_batch(None, None, None, ctypes.c_size_t(0), None, None, None) == -1
    assert lib.jfhip_field_op(9, 0, None, None, None, ctypes.c_size_t(1)) == -1
    assert lib.jfhip_phase_name(0, 2) == b"bucket_accumulate"


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under jellyfish_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "jellyfish_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("test oracle", ""), os.path.join(dirpath, f)
