//! Raw bindings to `libjfhip.so`; one declaration per function of `include/jfhip.h`, same order, same names.
//! The header is the documentation: each C declaration there cites the jellyfish interface it replaces.
//!
//! UNCOMPILED in the authoring image (no Rust toolchain); kept in step with the header by `tests/test_abi.py`.
//!
//! Conventions (jfhip.h): field elements are arkworks' in-memory Montgomery limbs (`Fp.0.0`), so `&[Fr]` is passed as
//! `*const u64` without conversion; a packed affine point is x limbs then y limbs; every function returns `JFHIP_OK`
//! or a negative status and leaves a message for `jfhip_last_error`.
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_void};

pub const JFHIP_OK: i32 = 0;
pub const JFHIP_ERR_INVALID: i32 = -1;
pub const JFHIP_ERR_DEVICE: i32 = -2;
pub const JFHIP_ERR_NO_DEVICE: i32 = -3;
pub const JFHIP_ERR_OOM: i32 = -4;

pub const JFHIP_BN254: i32 = 0;
pub const JFHIP_BLS12_381: i32 = 1;

/// Device-resident prover parameters (powers_of_g and their window multiples).
#[repr(C)]
pub struct jfhip_srs {
    _p: [u8; 0],
}
/// A HIP stream and its MSM / NTT workspace.
#[repr(C)]
pub struct jfhip_stream {
    _p: [u8; 0],
}

extern "C" {
    // ---- runtime ----
    pub fn jfhip_init(device: i32) -> i32;
    pub fn jfhip_shutdown() -> i32;
    pub fn jfhip_cache_clear() -> i32;
    pub fn jfhip_last_error(buf: *mut c_char, len: usize) -> i32;
    pub fn jfhip_device_count() -> i32;
    pub fn jfhip_stream_create(out: *mut *mut jfhip_stream) -> i32;
    pub fn jfhip_stream_destroy(s: *mut jfhip_stream) -> i32;
    pub fn jfhip_stream_release_workspace(s: *mut jfhip_stream) -> i32;
    pub fn jfhip_stream_sync(s: *mut jfhip_stream) -> i32;
    pub fn jfhip_stream_raw(s: *mut jfhip_stream) -> *mut c_void;
    pub fn jfhip_stream_wrap(hip_stream: *mut c_void, out: *mut *mut jfhip_stream) -> i32;

    // ---- SRS ----
    pub fn jfhip_srs_load(
        curve_id: i32,
        points: *const c_void,
        n: usize,
        stride_bytes: usize,
        window_bits: i32,
        out: *mut *mut jfhip_srs,
    ) -> i32;
    pub fn jfhip_srs_load_device(
        curve_id: i32,
        d_points_packed: *const c_void,
        n: usize,
        window_bits: i32,
        out: *mut *mut jfhip_srs,
    ) -> i32;
    /// Lagrange-basis commit key of the domain of size 2^log_n (+ `blinders` vanishing-polynomial points), a new handle.
    pub fn jfhip_srs_lagrange(
        srs: *const jfhip_srs,
        log_n: u32,
        blinders: u32,
        window_bits: i32,
        out: *mut *mut jfhip_srs,
    ) -> i32;
    pub fn jfhip_srs_load_serialized(
        curve_id: i32,
        bytes: *const u8,
        len: usize,
        n: usize,
        compressed: i32,
        validate: i32,
        window_bits: i32,
        out: *mut *mut jfhip_srs,
    ) -> i32;
    pub fn jfhip_srs_gen_for_testing(
        curve_id: i32,
        beta_mont: *const u64,
        g_xy: *const u64,
        n: usize,
        window_bits: i32,
        out: *mut *mut jfhip_srs,
    ) -> i32;
    pub fn jfhip_srs_release_caches(srs: *mut jfhip_srs) -> i32;
    pub fn jfhip_srs_free(srs: *mut jfhip_srs) -> i32;
    pub fn jfhip_srs_len(srs: *const jfhip_srs) -> usize;
    pub fn jfhip_srs_window_bits(srs: *const jfhip_srs) -> i32;
    pub fn jfhip_srs_read(srs: *const jfhip_srs, offset: usize, n: usize, out_points_packed: *mut c_void) -> i32;

    // ---- MSM / KZG commit ----
    pub fn jfhip_msm(
        srs: *const jfhip_srs,
        offset: usize,
        scalars_mont: *const u64,
        n: usize,
        out_xy: *mut c_void,
        out_infinity: *mut u8,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_batch(
        srs: *const jfhip_srs,
        scalars_mont: *const *const u64,
        lens: *const usize,
        batch: usize,
        out_xy: *mut c_void,
        out_infinity: *mut u8,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_batch_device(
        srs: *const jfhip_srs,
        d_scalars_mont: *const *const c_void,
        lens: *const usize,
        batch: usize,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_device(
        srs: *const jfhip_srs,
        offset: usize,
        d_scalars_mont: *const c_void,
        n: usize,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_small(
        curve_id: i32,
        points_xy: *const u64,
        scalars_mont: *const u64,
        n: usize,
        out_xy: *mut c_void,
        out_infinity: *mut u8,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_partial_device(
        srs: *const jfhip_srs,
        offset: usize,
        d_scalars_mont: *const c_void,
        n: usize,
        d_out_xyzz: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_msm_batch_partial_device(
        srs: *const jfhip_srs,
        offsets: *const usize,
        d_scalars_mont: *const *const c_void,
        lens: *const usize,
        batch: usize,
        d_out_xyzz: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_g1_sum_partials(
        curve_id: i32,
        d_partials_xyzz: *const c_void,
        count: usize,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_kzg_open(
        srs: *const jfhip_srs,
        coeffs_mont: *const u64,
        n: usize,
        point_mont: *const u64,
        out_proof_xy: *mut c_void,
        out_infinity: *mut u8,
        out_eval_mont: *mut u64,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_kzg_multi_point_open(
        srs: *const jfhip_srs,
        coeffs_mont: *const u64,
        n: usize,
        points_mont: *const u64,
        num_points: usize,
        out_proof_xy: *mut c_void,
        out_infinity: *mut u8,
        out_evals_mont: *mut u64,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_kzg_multi_open_rou(
        srs: *const jfhip_srs,
        coeffs_mont: *const u64,
        n: usize,
        log_domain: u32,
        num_points: usize,
        out_proofs_xy: *mut c_void,
        out_infinity: *mut u8,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_mle_open(
        levels: *const *const jfhip_srs,
        nv: usize,
        evals_mont: *const u64,
        point_mont: *const u64,
        out_proofs_xy: *mut c_void,
        out_infinity: *mut u8,
        out_eval_mont: *mut u64,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_mle_fold_device(
        curve_id: i32,
        d_f: *const c_void,
        half: usize,
        point_mont: *const u64,
        d_q: *mut c_void,
        d_r: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;

    // ---- NTT ----
    pub fn jfhip_ntt(
        curve_id: i32,
        data: *mut u64,
        log_n: u32,
        batch: usize,
        inverse: i32,
        coset_offset_mont: *const u64,
        in_len: usize,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_ntt_device(
        curve_id: i32,
        d_data: *mut c_void,
        log_n: u32,
        batch: usize,
        inverse: i32,
        coset_offset_mont: *const u64,
        in_len: usize,
        stream: *mut jfhip_stream,
    ) -> i32;

    // ---- PLONK quotient evaluations ----
    pub fn jfhip_plonk_quotient_turbo_device(
        curve_id: i32,
        log_n: u32,
        log_m: u32,
        d_selectors: *const c_void,
        d_sigmas: *const c_void,
        d_wires: *const c_void,
        d_prod_perm: *const c_void,
        d_pub_input: *const c_void,
        k_mont: *const u64,
        challenges_mont: *const u64,
        d_quot: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_quotient_ultra_device(
        curve_id: i32,
        log_n: u32,
        log_m: u32,
        d_selectors: *const c_void,
        d_sigmas: *const c_void,
        d_wires: *const c_void,
        d_prod_perm: *const c_void,
        d_pub_input: *const c_void,
        d_plookup: *const c_void,
        k_mont: *const u64,
        challenges_mont: *const u64,
        d_quot: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_quotient_coset_device(
        curve_id: i32,
        num_wire_types: u32,
        log_n: u32,
        log_m: u32,
        coset_offset_mont: *const u64,
        d_selectors: *const c_void,
        d_sigmas: *const c_void,
        d_wires: *const c_void,
        d_prod_perm: *const c_void,
        d_pub_input: *const c_void,
        d_plookup: *const c_void,
        k_mont: *const u64,
        challenges_mont: *const u64,
        d_quot: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_quotient_masked_device(
        curve_id: i32,
        num_wire_types: u32,
        log_n: u32,
        log_m: u32,
        coset_offset_mont: *const u64,
        d_selectors: *const c_void,
        d_sigmas: *const c_void,
        d_wires: *const c_void,
        d_prod_perm: *const c_void,
        d_pub_input: *const c_void,
        d_plookup: *const c_void,
        k_mont: *const u64,
        challenges_mont: *const u64,
        zero_selector_mask: u32,
        d_quot: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;

    // ---- grand-product builders ----
    pub fn jfhip_plonk_perm_product_device(
        curve_id: i32,
        log_n: u32,
        num_wire_types: u32,
        d_wire_vals: *const c_void,
        d_id_perm: *const c_void,
        d_sigma_vals: *const c_void,
        beta_mont: *const u64,
        gamma_mont: *const u64,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_lookup_product_device(
        curve_id: i32,
        log_n: u32,
        d_lookup_wire: *const c_void,
        d_table: *const c_void,
        d_sorted: *const c_void,
        beta_mont: *const u64,
        gamma_mont: *const u64,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_lookup_sorted_device(
        d_table: *const c_void,
        n_table: usize,
        d_lookups: *const c_void,
        n_lookups: usize,
        d_sorted: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_plonk_lookup_merge_device(
        curve_id: i32,
        log_n: u32,
        d_wire_vals: *const c_void,
        d_table_vals: *const c_void,
        tau_mont: *const u64,
        d_merged_table: *mut c_void,
        d_merged_lookup: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;

    // ---- device-resident polynomial glue ----
    pub fn jfhip_poly_divide_linear_device(
        curve_id: i32,
        d_coeffs: *const c_void,
        n: usize,
        point_mont: *const u64,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_poly_divide_linear_noalloc_device(
        curve_id: i32,
        d_coeffs: *const c_void,
        n: usize,
        point_mont: *const u64,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_poly_eval_batch_device(
        curve_id: i32,
        d_polys: *const *const c_void,
        lens: *const usize,
        count: usize,
        point_mont: *const u64,
        d_out: *mut c_void,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_poly_lincomb_device(
        curve_id: i32,
        d_polys: *const *const c_void,
        lens: *const usize,
        count: usize,
        coeffs_mont: *const u64,
        d_out: *mut c_void,
        n: usize,
        stream: *mut jfhip_stream,
    ) -> i32;
    pub fn jfhip_witness_gather_device(
        d_witness: *const c_void,
        num_vars: usize,
        d_wire_variables: *const u32,
        count: usize,
        d_out: *mut c_void,
        check: i32,
        stream: *mut jfhip_stream,
    ) -> i32;

    // ---- field / group vector ops, host utilities, probes ----
    pub fn jfhip_field_op(field_id: i32, op: i32, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> i32;
    pub fn jfhip_g1_op(curve_id: i32, op: i32, p: *const u64, q: *const u64, out: *mut u64) -> i32;
    pub fn jfhip_keccak_f1600(state25: *mut u64) -> i32;
    pub fn jfhip_probe(kind: i32, arg: i32, out_rate: *mut f64) -> i32;
    pub fn jfhip_marker_push(name: *const c_char) -> i32;
    pub fn jfhip_marker_pop() -> i32;
    pub fn jfhip_set_profiling(enabled: i32) -> i32;
    pub fn jfhip_last_phase_ms(stream: *mut jfhip_stream, phases: *mut f64, cap: i32) -> i32;
    pub fn jfhip_phase_name(kind: i32, idx: i32) -> *const c_char;
}
