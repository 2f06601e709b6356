/* TEST INFRASTRUCTURE ONLY -- CPU restatement, not product code.
 *
 * Short-Weierstrass G1 arithmetic (a = 0) over the base field instantiated as FQ(*)
 * with QL u64 limbs; scalars are 4-limb Fr.  Restates what the reference obtains from
 * ark-ec (external, arkworks 0.5): `E::G1::msm_bigint(bases, scalars).into_affine()`
 * at pcs/src/univariate_kzg/mod.rs:114-118 and the SRS definition
 * pcs/src/univariate_kzg/srs.rs:175-225.
 *
 * Packed affine point = x[QL] || y[QL] Montgomery limbs; the identity is x = y = 0
 * (not on either curve: b != 0).
 */
typedef struct { uint64_t X[QL], Y[QL], Z[QL]; } EC(jac);

static inline void EC(jac_set_inf)(const FQ(ctx) *f, EC(jac) *P) {
    FQ(copy)(P->X, f->r1);
    FQ(copy)(P->Y, f->r1);
    FQ(zero)(P->Z);
}
static inline int EC(jac_is_inf)(const EC(jac) *P) { return FQ(is_zero)(P->Z); }
static inline int EC(aff_is_inf)(const uint64_t *xy) { return FQ(is_zero)(xy) && FQ(is_zero)(xy + QL); }

static inline void EC(jac_from_aff)(const FQ(ctx) *f, EC(jac) *P, const uint64_t *xy) {
    if (EC(aff_is_inf)(xy)) { EC(jac_set_inf)(f, P); return; }
    FQ(copy)(P->X, xy);
    FQ(copy)(P->Y, xy + QL);
    FQ(copy)(P->Z, f->r1);
}

static inline void EC(jac_double)(const FQ(ctx) *f, EC(jac) *R, const EC(jac) *P) {
    if (EC(jac_is_inf)(P) || FQ(is_zero)(P->Y)) { EC(jac_set_inf)(f, R); return; }
    uint64_t A[QL], B[QL], C[QL], D[QL], E[QL], F[QL], t[QL], Z3[QL];
    FQ(sqr)(f, A, P->X);
    FQ(sqr)(f, B, P->Y);
    FQ(sqr)(f, C, B);
    FQ(add)(f, t, P->X, B);
    FQ(sqr)(f, t, t);
    FQ(sub)(f, t, t, A);
    FQ(sub)(f, t, t, C);
    FQ(dbl)(f, D, t);
    FQ(dbl)(f, E, A);
    FQ(add)(f, E, E, A);
    FQ(sqr)(f, F, E);
    FQ(mul)(f, Z3, P->Y, P->Z);
    FQ(dbl)(f, Z3, Z3);
    FQ(dbl)(f, t, D);
    FQ(sub)(f, R->X, F, t);
    FQ(sub)(f, t, D, R->X);
    FQ(mul)(f, t, E, t);
    FQ(dbl)(f, C, C);
    FQ(dbl)(f, C, C);
    FQ(dbl)(f, C, C);
    FQ(sub)(f, R->Y, t, C);
    FQ(copy)(R->Z, Z3);
}

static inline void EC(jac_add)(const FQ(ctx) *f, EC(jac) *R, const EC(jac) *P, const EC(jac) *Q) {
    if (EC(jac_is_inf)(P)) { *R = *Q; return; }
    if (EC(jac_is_inf)(Q)) { *R = *P; return; }
    uint64_t Z1Z1[QL], Z2Z2[QL], U1[QL], U2[QL], S1[QL], S2[QL], H[QL], Rr[QL], HH[QL], HHH[QL], V[QL], t[QL];
    FQ(sqr)(f, Z1Z1, P->Z);
    FQ(sqr)(f, Z2Z2, Q->Z);
    FQ(mul)(f, U1, P->X, Z2Z2);
    FQ(mul)(f, U2, Q->X, Z1Z1);
    FQ(mul)(f, S1, P->Y, Q->Z);
    FQ(mul)(f, S1, S1, Z2Z2);
    FQ(mul)(f, S2, Q->Y, P->Z);
    FQ(mul)(f, S2, S2, Z1Z1);
    if (FQ(eq)(U1, U2)) {
        if (FQ(eq)(S1, S2)) { EC(jac_double)(f, R, P); return; }
        EC(jac_set_inf)(f, R);
        return;
    }
    FQ(sub)(f, H, U2, U1);
    FQ(sub)(f, Rr, S2, S1);
    FQ(sqr)(f, HH, H);
    FQ(mul)(f, HHH, H, HH);
    FQ(mul)(f, V, U1, HH);
    EC(jac) O;
    FQ(sqr)(f, O.X, Rr);
    FQ(sub)(f, O.X, O.X, HHH);
    FQ(dbl)(f, t, V);
    FQ(sub)(f, O.X, O.X, t);
    FQ(sub)(f, t, V, O.X);
    FQ(mul)(f, t, Rr, t);
    FQ(mul)(f, S1, S1, HHH);
    FQ(sub)(f, O.Y, t, S1);
    FQ(mul)(f, O.Z, P->Z, Q->Z);
    FQ(mul)(f, O.Z, O.Z, H);
    *R = O;
}

/* R = P + (x,y) affine, negate the affine point when neg != 0 */
static inline void EC(jac_add_aff)(const FQ(ctx) *f, EC(jac) *R, const EC(jac) *P, const uint64_t *xy, int neg) {
    if (EC(aff_is_inf)(xy)) { *R = *P; return; }
    EC(jac) Q;
    FQ(copy)(Q.X, xy);
    if (neg) FQ(neg)(f, Q.Y, xy + QL); else FQ(copy)(Q.Y, xy + QL);
    FQ(copy)(Q.Z, f->r1);
    if (EC(jac_is_inf)(P)) { *R = Q; return; }
    uint64_t Z1Z1[QL], U2[QL], S2[QL], H[QL], Rr[QL], HH[QL], HHH[QL], V[QL], t[QL];
    FQ(sqr)(f, Z1Z1, P->Z);
    FQ(mul)(f, U2, Q.X, Z1Z1);
    FQ(mul)(f, S2, Q.Y, P->Z);
    FQ(mul)(f, S2, S2, Z1Z1);
    if (FQ(eq)(P->X, U2)) {
        if (FQ(eq)(P->Y, S2)) { EC(jac_double)(f, R, P); return; }
        EC(jac_set_inf)(f, R);
        return;
    }
    FQ(sub)(f, H, U2, P->X);
    FQ(sub)(f, Rr, S2, P->Y);
    FQ(sqr)(f, HH, H);
    FQ(mul)(f, HHH, H, HH);
    FQ(mul)(f, V, P->X, HH);
    EC(jac) O;
    FQ(sqr)(f, O.X, Rr);
    FQ(sub)(f, O.X, O.X, HHH);
    FQ(dbl)(f, t, V);
    FQ(sub)(f, O.X, O.X, t);
    FQ(sub)(f, t, V, O.X);
    FQ(mul)(f, t, Rr, t);
    FQ(mul)(f, S2, P->Y, HHH);
    FQ(sub)(f, O.Y, t, S2);
    FQ(mul)(f, O.Z, P->Z, H);
    *R = O;
}

static inline void EC(jac_to_aff)(const FQ(ctx) *f, uint64_t *xy, const EC(jac) *P) {
    if (EC(jac_is_inf)(P)) { FQ(zero)(xy); FQ(zero)(xy + QL); return; }
    uint64_t zi[QL], zi2[QL];
    FQ(inv)(f, zi, P->Z);
    FQ(sqr)(f, zi2, zi);
    FQ(mul)(f, xy, P->X, zi2);
    FQ(mul)(f, zi2, zi2, zi);
    FQ(mul)(f, xy + QL, P->Y, zi2);
}

/* k raw (non-Montgomery) little-endian limbs */
static inline void EC(jac_mul)(const FQ(ctx) *f, EC(jac) *R, const EC(jac) *P, const uint64_t *k, int nk) {
    EC(jac) acc, base = *P;
    EC(jac_set_inf)(f, &acc);
    for (int i = 0; i < nk; i++) {
        uint64_t w = k[i];
        for (int b = 0; b < 64; b++) {
            if (w & 1) EC(jac_add)(f, &acc, &acc, &base);
            EC(jac_double)(f, &base, &base);
            w >>= 1;
        }
    }
    *R = acc;
}

/* Batch normalisation (Montgomery's trick): jac[n] -> packed affine out[n][2*QL] */
static void EC(batch_to_aff)(const FQ(ctx) *f, uint64_t *out, const EC(jac) *pts, size_t n) {
    if (n == 0) return;
    uint64_t *pref = (uint64_t *)malloc(sizeof(uint64_t) * QL * n);
    uint64_t acc[QL];
    FQ(copy)(acc, f->r1);
    for (size_t i = 0; i < n; i++) {
        FQ(copy)(pref + i * QL, acc);
        if (!EC(jac_is_inf)(&pts[i])) FQ(mul)(f, acc, acc, pts[i].Z);
    }
    uint64_t inv[QL];
    FQ(inv)(f, inv, acc);
    for (size_t i = n; i-- > 0;) {
        uint64_t *xy = out + i * 2 * QL;
        if (EC(jac_is_inf)(&pts[i])) { FQ(zero)(xy); FQ(zero)(xy + QL); continue; }
        uint64_t zi[QL], zi2[QL];
        FQ(mul)(f, zi, inv, pref + i * QL);
        FQ(mul)(f, inv, inv, pts[i].Z);
        FQ(sqr)(f, zi2, zi);
        FQ(mul)(f, xy, pts[i].X, zi2);
        FQ(mul)(f, zi2, zi2, zi);
        FQ(mul)(f, xy + QL, pts[i].Y, zi2);
    }
    free(pref);
}

/* Windowed Pippenger, unsigned digits, window rule of ark-ec: c = 3 if n < 32 else ln(n)+2.
 * scalars: raw (already de-Montgomerised) 4-limb integers < r.  OpenMP over windows, the
 * same decomposition ark-ec parallelises with rayon. */
static void EC(msm_pippenger)(const FQ(ctx) *f, EC(jac) *result, const uint64_t *bases, const uint64_t *scal_raw,
                              size_t n, int scalar_bits, int threads) {
    int c;
    if (n < 32) c = 3;
    else {
        /* ln(n) ~ log2(n) * 69 / 100, as ark-std's ln_without_floats */
        int lg = 0;
        while (((size_t)1 << (lg + 1)) <= n) lg++;
        c = lg * 69 / 100 + 2;
    }
    int nwin = (scalar_bits + c - 1) / c;
    EC(jac) *wsum = (EC(jac) *)malloc(sizeof(EC(jac)) * nwin);
    /* when few windows but many threads, split the point range as well */
    int parts = 1;
    if (threads > nwin) parts = (threads + nwin - 1) / nwin;
    if ((size_t)parts > n / 1024 + 1) parts = (int)(n / 1024 + 1);
    EC(jac) *psum = (EC(jac) *)malloc(sizeof(EC(jac)) * nwin * parts);
    int tasks = nwin * parts;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int task = 0; task < tasks; task++) {
        int w = task / parts, part = task % parts;
        size_t lo = n * part / parts, hi = n * (part + 1) / parts;
        size_t nb = ((size_t)1 << c) - 1;
        EC(jac) *buckets = (EC(jac) *)malloc(sizeof(EC(jac)) * nb);
        for (size_t b = 0; b < nb; b++) EC(jac_set_inf)(f, &buckets[b]);
        int bit0 = w * c;
        for (size_t i = lo; i < hi; i++) {
            const uint64_t *s = scal_raw + 4 * i;
            int limb = bit0 >> 6, sh = bit0 & 63;
            uint64_t d = s[limb] >> sh;
            if (sh + c > 64 && limb + 1 < 4) d |= s[limb + 1] << (64 - sh);
            d &= ((uint64_t)1 << c) - 1;
            if (d) EC(jac_add_aff)(f, &buckets[d - 1], &buckets[d - 1], bases + i * 2 * QL, 0);
        }
        EC(jac) run, sum;
        EC(jac_set_inf)(f, &run);
        EC(jac_set_inf)(f, &sum);
        for (size_t b = nb; b-- > 0;) {
            EC(jac_add)(f, &run, &run, &buckets[b]);
            EC(jac_add)(f, &sum, &sum, &run);
        }
        psum[task] = sum;
        free(buckets);
    }
    for (int w = 0; w < nwin; w++) {
        EC(jac_set_inf)(f, &wsum[w]);
        for (int q = 0; q < parts; q++) EC(jac_add)(f, &wsum[w], &wsum[w], &psum[w * parts + q]);
    }
    EC(jac) acc;
    EC(jac_set_inf)(f, &acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) EC(jac_double)(f, &acc, &acc);
        EC(jac_add)(f, &acc, &acc, &wsum[w]);
    }
    *result = acc;
    free(wsum);
    free(psum);
}
