/* TEST INFRASTRUCTURE ONLY -- CPU restatement, not product code.
 *
 * Montgomery prime-field arithmetic on NL little-endian u64 limbs, R = 2^(64*NL),
 * the in-memory form arkworks' Fp<MontBackend<_, NL>> holds in `.0.0`
 * (reference: pcs/src/univariate_kzg/mod.rs:970-979 copies exactly these limbs).
 * Instantiated by defining NL and PFX(name) before including this file.
 */
typedef struct {
    uint64_t p[NL];    /* modulus */
    uint64_t r1[NL];   /* R mod p  (Montgomery one) */
    uint64_t r2[NL];   /* R^2 mod p */
    uint64_t inv;      /* -p^{-1} mod 2^64 */
} PFX(ctx);

static inline int PFX(geq)(const uint64_t *a, const uint64_t *b) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline int PFX(is_zero)(const uint64_t *a) {
    uint64_t t = 0;
    for (int i = 0; i < NL; i++) t |= a[i];
    return t == 0;
}
static inline int PFX(eq)(const uint64_t *a, const uint64_t *b) {
    uint64_t t = 0;
    for (int i = 0; i < NL; i++) t |= a[i] ^ b[i];
    return t == 0;
}
static inline void PFX(copy)(uint64_t *r, const uint64_t *a) {
    for (int i = 0; i < NL; i++) r[i] = a[i];
}
static inline void PFX(zero)(uint64_t *r) {
    for (int i = 0; i < NL; i++) r[i] = 0;
}
static inline uint64_t PFX(sub_raw)(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    unsigned __int128 br = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
        r[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    return (uint64_t)br;
}
static inline uint64_t PFX(add_raw)(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    unsigned __int128 c = 0;
    for (int i = 0; i < NL; i++) {
        c += (unsigned __int128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static inline void PFX(add)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t t[NL];
    uint64_t c = PFX(add_raw)(t, a, b);
    if (c || PFX(geq)(t, f->p)) PFX(sub_raw)(t, t, f->p);
    PFX(copy)(r, t);
}
static inline void PFX(sub)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t t[NL];
    if (PFX(sub_raw)(t, a, b)) PFX(add_raw)(t, t, f->p);
    PFX(copy)(r, t);
}
static inline void PFX(neg)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) {
    if (PFX(is_zero)(a)) { PFX(zero)(r); return; }
    uint64_t t[NL];
    PFX(sub_raw)(t, f->p, a);
    PFX(copy)(r, t);
}
static inline void PFX(dbl)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) { PFX(add)(f, r, a, a); }

/* CIOS Montgomery product: r = a*b*R^{-1} mod p, fully reduced. */
static inline void PFX(mul)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a, const uint64_t *b) {
    uint64_t t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL] = (uint64_t)c;
        t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * f->inv;
        c = (unsigned __int128)m * f->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (unsigned __int128)m * f->p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    if (t[NL] || PFX(geq)(t, f->p)) PFX(sub_raw)(t, t, f->p);
    PFX(copy)(r, t);
}
static inline void PFX(sqr)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) { PFX(mul)(f, r, a, a); }

static inline void PFX(from_mont)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) {
    uint64_t one[NL];
    PFX(zero)(one);
    one[0] = 1;
    PFX(mul)(f, r, a, one);
}
static inline void PFX(to_mont)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) { PFX(mul)(f, r, a, f->r2); }

/* r = a^e (e raw little-endian limbs, ne of them) */
static inline void PFX(pow)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a, const uint64_t *e, int ne) {
    uint64_t acc[NL], base[NL];
    PFX(copy)(acc, f->r1);
    PFX(copy)(base, a);
    for (int i = 0; i < ne; i++) {
        uint64_t w = e[i];
        for (int b = 0; b < 64; b++) {
            if (w & 1) PFX(mul)(f, acc, acc, base);
            PFX(mul)(f, base, base, base);
            w >>= 1;
        }
    }
    PFX(copy)(r, acc);
}
/* Fermat inverse; inv(0) = 0 */
static inline void PFX(inv)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a) {
    uint64_t e[NL], two[NL];
    PFX(zero)(two);
    two[0] = 2;
    PFX(sub_raw)(e, f->p, two);
    PFX(pow)(f, r, a, e, NL);
}
static inline void PFX(pow_u64)(const PFX(ctx) *f, uint64_t *r, const uint64_t *a, uint64_t e) {
    PFX(pow)(f, r, a, &e, 1);
}
/* small integer -> Montgomery */
static inline void PFX(from_u64)(const PFX(ctx) *f, uint64_t *r, uint64_t v) {
    uint64_t t[NL];
    PFX(zero)(t);
    t[0] = v;
    PFX(to_mont)(f, r, t);
}
