//! NTT seam of the `hip` feature: drop-in replacements for the `Radix2EvaluationDomain::{fft, ifft, ifft_in_place}` and
//! `get_coset(offset).{fft, ifft}` calls of the PLONK prover (reference plonk/src/proof_system/prover.rs:534-573,
//! 658-660; relation/src/constraint_system.rs:1369, 1386, 1418, 1437, 1454, 1464, 1473, 1482, 1491, 1573, 1621-1622),
//! routed to `jfhip_ntt` (include/jfhip.h) for the scalar fields the library implements and to ark-poly otherwise.
//!
//! UNCOMPILED: the authoring image has no Rust toolchain.  The same calls are compiled and tested from C++
//! (include/jf_pcs.hpp `Radix2EvaluationDomain`, tests/cpp/test_pcs.cpp) and Python (jellyfish_amd/lib.py).
//!
//! arkworks' `Fp<MontBackend<_, 4>, 4>` is four little-endian u64 limbs in Montgomery form, which is what the library
//! transforms in place, so `&mut [F]` is passed by pointer (size asserted below).
#![cfg_attr(not(test), no_std)]
extern crate alloc;

use alloc::{vec, vec::Vec};
use ark_ff::FftField;
use ark_poly::{EvaluationDomain, GeneralEvaluationDomain, Radix2EvaluationDomain};
use core::{any::TypeId, mem::size_of, ptr};
use jf_hip_sys::*;

/// `curve_id` of jfhip.h for a scalar field the library implements
fn curve_of<F: FftField>() -> Option<i32> {
    let t = TypeId::of::<F>();
    if size_of::<F>() != 32 {
        return None;
    }
    if t == TypeId::of::<ark_bn254::Fr>() {
        Some(JFHIP_BN254)
    } else if t == TypeId::of::<ark_bls12_381::Fr>() {
        Some(JFHIP_BLS12_381)
    } else {
        None
    }
}

/// Domains below this size stay on the host: the call is dominated by the PCIe round trip.
pub const MIN_DEVICE_LOG_SIZE: u64 = 12;

/// `data`: `batch` polynomials of `2^log_n` elements each, back to back; only the first `in_len` of each are read.
/// Returns false when the device did not take the call (unsupported field, small domain, library error): the caller
/// then runs ark-poly on the same buffer, which is untouched in that case.
fn device_ntt<F: FftField>(
    data: &mut [F],
    log_n: u64,
    batch: usize,
    inverse: bool,
    offset: Option<F>,
    in_len: usize,
) -> bool {
    let Some(id) = curve_of::<F>() else { return false };
    if log_n < MIN_DEVICE_LOG_SIZE || data.len() != batch << log_n {
        return false;
    }
    let off_ptr = match offset.as_ref() {
        Some(h) if !h.is_one() => h as *const F as *const u64,
        _ => ptr::null(),
    };
    // the library transforms a staging copy on the device and writes `data` only on success
    let rc = unsafe {
        jfhip_ntt(
            id,
            data.as_mut_ptr() as *mut u64,
            log_n as u32,
            batch,
            inverse as i32,
            off_ptr,
            in_len,
            ptr::null_mut(),
        )
    };
    rc == JFHIP_OK
}

/// `domain.ifft_in_place(v)` (constraint_system.rs:1418, 1437, 1454, 1573): zero-extends `v` to the domain size.
pub fn ifft_in_place<F: FftField>(domain: &Radix2EvaluationDomain<F>, v: &mut Vec<F>) {
    let in_len = v.len();
    if in_len <= domain.size() {
        v.resize(domain.size(), F::zero());
        if device_ntt(v, domain.log_size_of_group as u64, 1, true, Some(domain.coset_offset()), in_len) {
            return;
        }
        v.truncate(in_len);
    }
    domain.ifft_in_place(v)
}

/// `domain.ifft(evals)` (constraint_system.rs:1369, 1386, 1464, 1473, 1482, 1491, 1621-1622)
pub fn ifft<F: FftField>(domain: &Radix2EvaluationDomain<F>, evals: &[F]) -> Vec<F> {
    let mut v = evals.to_vec();
    ifft_in_place(domain, &mut v);
    v
}

/// The selector / sigma / wire / Plookup `coset.fft(poly.coeffs())` calls of one proving instance
/// (prover.rs:534-573) as ONE batched transform: `polys[k]` -> its evaluations over the coset.
pub fn coset_fft_batch<F: FftField>(coset: &GeneralEvaluationDomain<F>, polys: &[&[F]]) -> Vec<Vec<F>> {
    if let GeneralEvaluationDomain::Radix2(d) = coset {
        let m = d.size();
        let in_len = polys.iter().map(|p| p.len()).max().unwrap_or(0);
        if in_len <= m && !polys.is_empty() {
            let mut buf = vec![F::zero(); polys.len() * m];
            for (k, p) in polys.iter().enumerate() {
                buf[k * m..k * m + p.len()].copy_from_slice(p);
            }
            if device_ntt(&mut buf, d.log_size_of_group as u64, polys.len(), false, Some(d.coset_offset()), in_len.max(1)) {
                return buf.chunks(m).map(|c| c.to_vec()).collect();
            }
        }
    }
    polys.iter().map(|p| coset.fft(p)).collect()
}

/// `coset.ifft(&quot_poly_coset_evals_sum)` (prover.rs:658-660)
pub fn coset_ifft<F: FftField>(coset: &GeneralEvaluationDomain<F>, evals: &[F]) -> Vec<F> {
    if let GeneralEvaluationDomain::Radix2(d) = coset {
        if evals.len() == d.size() {
            let mut v = evals.to_vec();
            if device_ntt(&mut v, d.log_size_of_group as u64, 1, true, Some(d.coset_offset()), evals.len()) {
                return v;
            }
        }
    }
    coset.ifft(evals)
}

#[cfg(test)]
mod tests {
    use super::*;
    use ark_std::UniformRand;

    #[test]
    fn layout() {
        assert_eq!(size_of::<ark_bn254::Fr>(), 32);
        assert_eq!(size_of::<ark_bls12_381::Fr>(), 32);
    }

    /// the seam returns what ark-poly returns (the protocol of check_polynomial, constraint_system.rs:2262-2268)
    #[test]
    fn device_ntt_equals_ark_poly() {
        let mut rng = ark_std::test_rng();
        type F = ark_bls12_381::Fr;
        let domain = Radix2EvaluationDomain::<F>::new(1 << 13).unwrap();
        let evals: Vec<F> = (0..(1 << 13) - 5).map(|_| F::rand(&mut rng)).collect();
        assert_eq!(ifft(&domain, &evals), domain.ifft(&evals));
        let coset = GeneralEvaluationDomain::<F>::new(1 << 16).unwrap().get_coset(F::GENERATOR).unwrap();
        let polys: Vec<Vec<F>> = (0..3).map(|k| (0..(1 << 13) + k).map(|_| F::rand(&mut rng)).collect()).collect();
        let refs: Vec<&[F]> = polys.iter().map(|p| p.as_slice()).collect();
        let got = coset_fft_batch(&coset, &refs);
        for (g, p) in got.iter().zip(polys.iter()) {
            assert_eq!(g, &coset.fft(p));
        }
        assert_eq!(coset_ifft(&coset, &got[0])[..polys[0].len()], polys[0][..]);
    }
}
