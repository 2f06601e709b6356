// C++ parity test of the host-side mirror (include/jf_pcs.hpp), reading like the reference's own
// test_gpu_e2e (pcs/src/univariate_kzg/mod.rs:1359-1414): gpu_commit == commit (the CPU side here is
// the oracle library), batches of 10..14, the empty batch, open(), error behaviour, and the
// evaluation-domain round trip of check_polynomial (relation/src/constraint_system.rs:2262-2268).
// TEST CODE: links the oracle (libjforacle.so) as the checker.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../include/jf_pcs.hpp"

extern "C" {
int jfo_srs_gen(int curve, const uint64_t* beta_mont, const uint64_t* g_xy, size_t n, uint64_t* out, int threads);
int jfo_msm(int curve, const uint64_t* bases, const uint64_t* scalars_mont, size_t n, uint64_t* out_xy, int threads);
int jfo_open_witness(int curve, const uint64_t* coeffs, size_t n, const uint64_t* z, uint64_t* q_out, uint64_t* eval_out);
int jfo_poly_eval(int curve, const uint64_t* coeffs, size_t n, const uint64_t* z, uint64_t* out);
int jfo_ntt(int curve, uint64_t* data, int log_n, size_t batch, int inverse, const uint64_t* coset_mont, int threads);
int jfo_fp_unop(int field_id, int op, const uint64_t* a, uint64_t* out, size_t n);
int jfo_fp_binop(int field_id, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
}

using namespace jf_pcs;

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);  \
            std::exit(1);                                               \
        }                                                               \
    } while (0)

static std::mt19937_64 rng(0x5eed);

static Fr rand_fr() {
    Fr f;
    for (auto& l : f) l = rng();
    f[3] &= (1ull << 60) - 1;  // < r for both curves; any value < r is a valid Montgomery element
    return f;
}

template <class E>
static G1Affine<E> from_packed(const uint64_t* xy) {
    G1Affine<E> p;
    bool zero = true;
    for (int i = 0; i < 2 * E::FQ_LIMBS; i++) zero &= xy[i] == 0;
    if (zero) return G1Affine<E>::zero();
    std::memcpy(p.x.data(), xy, sizeof(p.x));
    std::memcpy(p.y.data(), xy + E::FQ_LIMBS, sizeof(p.y));
    p.infinity = false;
    return p;
}

template <class E>
static void run(const uint64_t* g_xy) {
    using PCS = UnivariateKzgPCS<E>;
    constexpr int PT = 2 * E::FQ_LIMBS;
    const size_t supported_degree = 1 << 12;
    // gen_srs_for_testing
    Fr beta = rand_fr();
    std::vector<uint64_t> packed((supported_degree + 1) * PT);
    jfo_srs_gen(E::ID, beta.data(), g_xy, supported_degree + 1, packed.data(), 8);
    UnivariateProverParam<E> pp;
    for (size_t i = 0; i <= supported_degree; i++) pp.powers_of_g.push_back(from_packed<E>(&packed[i * PT]));

    auto cpu_commit = [&](const DensePolynomial& p) {
        std::vector<uint64_t> out(PT);
        jfo_msm(E::ID, packed.data(), p.coeffs.empty() ? nullptr : p.coeffs[0].data(), p.coeffs.size(), out.data(), 8);
        return Commitment<E>{from_packed<E>(out.data())};
    };
    auto rand_poly = [&](size_t degree) {
        std::vector<Fr> c(degree + 1);
        for (auto& x : c) x = rand_fr();
        return DensePolynomial::from_coefficients_vec(c);
    };

    auto srs = PCS::load_prover_param_to_gpu(pp, supported_degree);
    CHECK(srs.is_ok());
    auto stream = warmup_new_stream();
    CHECK(stream.is_ok());

    // 10 rounds, random degree < 1025
    for (int round = 0; round < 10; round++) {
        size_t degree = 1 + rng() % 1024;
        DensePolynomial p = rand_poly(degree);
        auto comm_gpu = PCS::gpu_commit(pp, p);
        CHECK(comm_gpu.is_ok());
        CHECK(comm_gpu.unwrap() == cpu_commit(p));
        auto comm2 = PCS::gpu_commit_with_loaded_prover_param(srs.unwrap(), p, stream.unwrap());
        CHECK(comm2.is_ok() && comm2.unwrap() == comm_gpu.unwrap());
        auto comm3 = PCS::commit(srs.unwrap(), p);
        CHECK(comm3.is_ok() && comm3.unwrap() == comm_gpu.unwrap());
        // open: proof and evaluation against the oracle's synthetic division
        Fr z = rand_fr();
        auto op = PCS::open(srs.unwrap(), p, z);
        CHECK(op.is_ok());
        std::vector<Fr> q(degree);
        Fr ev;
        jfo_open_witness(E::ID, p.coeffs[0].data(), degree + 1, z.data(), q[0].data(), ev.data());
        CHECK(op.unwrap().second == ev);
        CHECK(op.unwrap().first.proof == cpu_commit(DensePolynomial{q}).point);
    }
    // batches of 10..14
    for (size_t bs = 10; bs < 15; bs++) {
        std::vector<DensePolynomial> polys;
        for (size_t i = 0; i < bs; i++) polys.push_back(rand_poly(1 + rng() % 1024));
        auto comms = PCS::gpu_batch_commit(pp, polys);
        CHECK(comms.is_ok() && comms.unwrap().size() == bs);
        for (size_t i = 0; i < bs; i++) CHECK(comms.unwrap()[i] == cpu_commit(polys[i]));
        auto comms2 = PCS::batch_commit(srs.unwrap(), polys);
        CHECK(comms2.is_ok());
        for (size_t i = 0; i < bs; i++) CHECK(comms2.unwrap()[i] == comms.unwrap()[i]);
    }
    // the trait's signatures (host prover parameters): one cached upload per powers_of_g allocation
    {
        PCS::evict_prover_param(pp);
        const size_t before = PCS::cached_prover_params();
        DensePolynomial p = rand_poly(700);
        auto c1 = PCS::commit(pp, p);
        CHECK(c1.is_ok() && c1.unwrap() == cpu_commit(p));
        CHECK(PCS::cached_prover_params() == before + 1);
        auto c2 = PCS::commit(pp, p);
        CHECK(c2.is_ok() && c2.unwrap() == c1.unwrap());
        CHECK(PCS::cached_prover_params() == before + 1);
        std::vector<DensePolynomial> polys{rand_poly(5), rand_poly(900), p};
        auto bc = PCS::batch_commit(pp, polys);
        CHECK(bc.is_ok() && bc.unwrap().size() == 3 && bc.unwrap()[2] == c1.unwrap());
        Fr z = rand_fr();
        auto o1 = PCS::open(pp, p, z);
        auto o2 = PCS::open(srs.unwrap(), p, z);
        CHECK(o1.is_ok() && o2.is_ok() && o1.unwrap().second == o2.unwrap().second);
        CHECK(o1.unwrap().first.proof == o2.unwrap().first.proof);
        auto e = PCS::commit(pp, rand_poly(supported_degree + 5));
        CHECK(e.is_err() && e.unwrap_err().msg.find("larger than allowed") != std::string::npos);
        // a prefix view of the same allocation is a different key (different length)
        UnivariateProverParam<E> half;
        half.powers_of_g.assign(pp.powers_of_g.begin(), pp.powers_of_g.begin() + 2049);
        auto c3 = PCS::commit(half, p);
        CHECK(c3.is_ok() && c3.unwrap() == c1.unwrap());
        CHECK(PCS::cached_prover_params() == before + 2);
        PCS::evict_prover_param(half);
        PCS::evict_prover_param(pp);
        CHECK(PCS::cached_prover_params() == before);
    }
    // a freed SRS whose address and length are reused by DIFFERENT parameters must not commit against the stale device
    // copy: the cache key carries a fingerprint of the contents (no evict call here, as after a plain drop + realloc)
    {
        UnivariateProverParam<E> reused;
        reused.powers_of_g.assign(pp.powers_of_g.begin(), pp.powers_of_g.begin() + 1025);
        DensePolynomial p = rand_poly(1000);
        auto c1 = PCS::commit(reused, p);
        CHECK(c1.is_ok() && c1.unwrap() == cpu_commit(p));
        const size_t held = PCS::cached_prover_params();
        // other parameters placed into the same buffer: [beta'^i] g with another beta
        Fr beta2 = rand_fr();
        std::vector<uint64_t> packed2(1025 * PT);
        jfo_srs_gen(E::ID, beta2.data(), g_xy, 1025, packed2.data(), 8);
        const void* addr = reused.powers_of_g.data();
        for (size_t i = 0; i < 1025; i++) reused.powers_of_g[i] = from_packed<E>(&packed2[i * PT]);
        CHECK(reused.powers_of_g.data() == addr);
        auto c2 = PCS::commit(reused, p);
        std::vector<uint64_t> out(PT);
        jfo_msm(E::ID, packed2.data(), p.coeffs[0].data(), p.coeffs.size(), out.data(), 8);
        CHECK(c2.is_ok() && c2.unwrap().point == from_packed<E>(out.data()));
        CHECK(!(c2.unwrap() == c1.unwrap()));
        CHECK(PCS::cached_prover_params() == held);   // the stale entry was replaced, not kept beside the new one
        PCS::evict_prover_param(reused);
    }
    // multi_point_open (:416-468): k successive synthetic divisions on the oracle side
    {
        DensePolynomial p = rand_poly(600);
        std::vector<Fr> pts{rand_fr(), rand_fr(), rand_fr()};
        auto mo = PCS::multi_point_open(srs.unwrap(), p, pts);
        CHECK(mo.is_ok() && mo.unwrap().second.size() == 3);
        std::vector<Fr> cur = p.coeffs;
        for (size_t i = 0; i < pts.size(); i++) {
            Fr ev;
            jfo_poly_eval(E::ID, p.coeffs[0].data(), p.coeffs.size(), pts[i].data(), ev.data());
            CHECK(mo.unwrap().second[i] == ev);
            std::vector<Fr> q(cur.size() - 1);
            Fr rem;
            jfo_open_witness(E::ID, cur[0].data(), cur.size(), pts[i].data(), q[0].data(), rem.data());
            cur = q;
        }
        CHECK(mo.unwrap().first.proof == cpu_commit(DensePolynomial{cur}).point);
        auto mo1 = PCS::multi_point_open(pp, p, {pts[0]});
        auto o1 = PCS::open(srs.unwrap(), p, pts[0]);
        CHECK(mo1.is_ok() && o1.is_ok() && mo1.unwrap().first.proof == o1.unwrap().first.proof);
        auto bad = PCS::multi_point_open(srs.unwrap(), p, {});
        CHECK(bad.is_err() && bad.unwrap_err().msg.find("no point to evaluate and open") != std::string::npos);
        PCS::evict_prover_param(pp);
    }
    // empty batch -> Ok(vec![])
    CHECK(PCS::gpu_batch_commit(pp, {}).is_ok());
    CHECK(PCS::gpu_batch_commit(pp, {}).unwrap().empty());
    // zero polynomial -> identity
    auto zc = PCS::commit(srs.unwrap(), DensePolynomial::from_coefficients_vec({Fr{0, 0, 0, 0}}));
    CHECK(zc.is_ok() && zc.unwrap().point.infinity);
    // low-order zero coefficients are skipped (offset into the SRS)
    {
        DensePolynomial p = rand_poly(300);
        for (int i = 0; i < 7; i++) p.coeffs[i] = Fr{0, 0, 0, 0};
        auto c = PCS::commit(srs.unwrap(), p);
        CHECK(c.is_ok() && c.unwrap() == cpu_commit(p));
    }
    // errors: degree too large, supported degree too large, mismatched batch_open
    {
        auto e = PCS::commit(srs.unwrap(), rand_poly(supported_degree + 5));
        CHECK(e.is_err() && e.unwrap_err().kind == PCSError::InvalidParameters);
        CHECK(e.unwrap_err().msg.find("larger than allowed") != std::string::npos);
        auto e2 = PCS::load_prover_param_to_gpu(pp, supported_degree + 1);
        CHECK(e2.is_err() && e2.unwrap_err().msg.find("supported degree") != std::string::npos);
        auto e3 = PCS::batch_open(srs.unwrap(), {rand_poly(3)}, {});
        CHECK(e3.is_err() && e3.unwrap_err().kind == PCSError::InvalidParameters);
    }
    // Lagrange-basis key: commit over evaluations (+ zero blinders) == the oracle's commit to the iFFT's coefficients
    {
        auto lag = srs.unwrap().lagrange(10, 3);
        CHECK(lag.is_ok() && lag.unwrap().len() == 1024 + 3);
        std::vector<Fr> evals(1024);
        for (auto& x : evals) x = rand_fr();
        std::vector<Fr> coeffs = evals;
        jfo_ntt(E::ID, coeffs[0].data(), 10, 1, 1, nullptr, 4);
        std::vector<Fr> sc = evals;
        sc.resize(1024 + 3);   // b_0 = b_1 = b_2 = 0
        auto c = PCS::gpu_commit_with_loaded_prover_param(lag.unwrap(), DensePolynomial{sc}, stream.unwrap());
        CHECK(c.is_ok() && c.unwrap() == cpu_commit(DensePolynomial{coeffs}));
        CHECK(srs.unwrap().lagrange(13, 0).is_err());   // 2^13 points: more than the key holds
    }
    // evaluation domain: ifft -> fft round trip, coset, against the oracle
    {
        auto dom = Radix2EvaluationDomain<E>::create(1000);
        CHECK(dom.is_ok() && dom.unwrap().size() == 1024);
        std::vector<Fr> evals(1024);
        for (auto& x : evals) x = rand_fr();
        auto coeffs = dom.unwrap().ifft(evals);
        CHECK(coeffs.is_ok());
        std::vector<Fr> ref = evals;
        jfo_ntt(E::ID, ref[0].data(), 10, 1, 1, nullptr, 4);
        CHECK(coeffs.unwrap() == ref);
        auto back = dom.unwrap().fft(coeffs.unwrap());
        CHECK(back.is_ok() && back.unwrap() == evals);
        // coset = GENERATOR, short input (zero-extended)
        uint64_t raw[4] = {E::GENERATOR, 0, 0, 0};
        Fr h;
        jfo_fp_unop(E::ID == 0 ? 0 : 2, 4, raw, h.data(), 1);
        auto coset = dom.unwrap().get_coset(h);
        std::vector<Fr> shortc(131);
        for (auto& x : shortc) x = rand_fr();
        auto ce = coset.fft(shortc);
        CHECK(ce.is_ok() && ce.unwrap().size() == 1024);
        std::vector<Fr> ref2(1024, Fr{0, 0, 0, 0});
        std::copy(shortc.begin(), shortc.end(), ref2.begin());
        jfo_ntt(E::ID, ref2[0].data(), 10, 1, 0, h.data(), 4);
        CHECK(ce.unwrap() == ref2);
        auto cb = coset.ifft(ce.unwrap());
        CHECK(cb.is_ok());
        for (size_t i = 0; i < 131; i++) CHECK(cb.unwrap()[i] == shortc[i]);
        CHECK(Radix2EvaluationDomain<E>::create((size_t)1 << 33).is_err());
    }
    // multi_open_rou_proofs == per-point open (the reference's own assertion, mod.rs:1237-1301)
    {
        DensePolynomial p = rand_poly(37);
        const uint32_t log_domain = 6;
        auto proofs = PCS::multi_open_rou_proofs(srs.unwrap(), p, 40, log_domain);
        CHECK(proofs.is_ok() && proofs.unwrap().size() == 40);
        auto dom = Radix2EvaluationDomain<E>::create((size_t)1 << log_domain);
        std::vector<Fr> unit(2, Fr{0, 0, 0, 0});
        const uint64_t one_bn[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
        const uint64_t one_bls[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
        std::memcpy(unit[1].data(), E::ID == 0 ? one_bn : one_bls, 32);   // the polynomial X: its evaluations are w^j
        auto ws = dom.unwrap().fft(unit);
        CHECK(ws.is_ok());
        for (size_t j : {(size_t)0, (size_t)1, (size_t)17, (size_t)39}) {
            auto op = PCS::open(srs.unwrap(), p, ws.unwrap()[j]);
            CHECK(op.is_ok() && op.unwrap().first.proof == proofs.unwrap()[j].proof);
        }
        auto bad = PCS::multi_open_rou_proofs(srs.unwrap(), rand_poly(100), 4, 5);
        CHECK(bad.is_err() && bad.unwrap_err().msg.find("smaller domain") != std::string::npos);
    }
    // MultilinearKzgPCS::commit / open (pcs/src/multilinear_kzg/mod.rs:123-149, 288-358).  The levels are ranges of the
    // generated points: the MSM sites see nothing else of the parameter.
    {
        using ML = MultilinearKzgPCS<E>;
        const size_t num_vars = 8;
        MultilinearProverParam<E> mpp;
        mpp.num_vars = num_vars;
        std::vector<size_t> level_at;
        size_t at = 0;
        for (size_t i = 0; i <= num_vars; i++) {
            const size_t sz = (size_t)1 << (num_vars - i);
            level_at.push_back(at);
            mpp.powers_of_g.emplace_back(pp.powers_of_g.begin() + at, pp.powers_of_g.begin() + at + sz);
            at += sz;
        }
        auto dpp = ML::load_prover_param_to_gpu(mpp);
        CHECK(dpp.is_ok());
        const int fid = E::ID == 0 ? 0 : 2;
        for (size_t nv : {(size_t)8, (size_t)5}) {
            DenseMultilinearExtension poly;
            poly.num_vars = nv;
            for (size_t i = 0; i < ((size_t)1 << nv); i++) poly.evaluations.push_back(rand_fr());
            std::vector<Fr> point(nv);
            for (auto& z : point) z = rand_fr();
            auto com = ML::commit(*dpp.unwrap(), poly);
            CHECK(com.is_ok());
            std::vector<uint64_t> out(PT);
            jfo_msm(E::ID, &packed[level_at[num_vars - nv] * PT], poly.evaluations[0].data(), poly.evaluations.size(),
                    out.data(), 8);
            CHECK(com.unwrap().point == from_packed<E>(out.data()));
            auto op = ML::open(*dpp.unwrap(), poly, point);
            CHECK(op.is_ok() && op.unwrap().first.proofs.size() == nv);
            std::vector<Fr> f = poly.evaluations;
            const size_t ignored = num_vars - nv + 1;
            for (size_t i = 0; i < nv; i++) {
                const size_t half = f.size() / 2;
                std::vector<Fr> q(half), r(half);
                for (size_t b = 0; b < half; b++) {
                    Fr t;
                    jfo_fp_binop(fid, 1, f[2 * b + 1].data(), f[2 * b].data(), q[b].data(), 1);
                    jfo_fp_binop(fid, 2, q[b].data(), point[i].data(), t.data(), 1);
                    jfo_fp_binop(fid, 0, f[2 * b].data(), t.data(), r[b].data(), 1);
                }
                jfo_msm(E::ID, &packed[level_at[ignored + i] * PT], q[0].data(), half, out.data(), 8);
                CHECK(op.unwrap().first.proofs[i] == from_packed<E>(out.data()));
                f = r;
            }
            CHECK(op.unwrap().second == f[0]);
        }
        DenseMultilinearExtension big;
        big.num_vars = num_vars + 1;
        big.evaluations.assign((size_t)2 << num_vars, rand_fr());
        auto e1 = ML::commit(*dpp.unwrap(), big);
        CHECK(e1.is_err() && e1.unwrap_err().msg.find("exceeds param limit") != std::string::npos);
        DenseMultilinearExtension p3;
        p3.num_vars = 3;
        p3.evaluations.assign(8, rand_fr());
        auto e2 = ML::open(*dpp.unwrap(), p3, std::vector<Fr>(2, rand_fr()));
        CHECK(e2.is_err() && e2.unwrap_err().msg.find("does not match point len") != std::string::npos);
    }
    std::printf("%s ok\n", E::ID == 0 ? "bn254" : "bls12_381");
}

int main() {
    if (jfhip_init(0) != JFHIP_OK) {
        char buf[300];
        jfhip_last_error(buf, sizeof(buf));
        std::printf("no GPU: %s\n", buf);
        return 2;
    }
    // generators in Montgomery form
    const uint64_t bn_g[8] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full,
                              0xa6ba871b8b1e1b3aull, 0x14f1d651eb8e167bull, 0xccdd46def0f28c58ull, 0x1c14ef83340fbe5eull};
    const uint64_t bls_g[12] = {0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull, 0xf0ae6acdf3d0e747ull,
                                0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull, 0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull,
                                0xdd595f13570725ceull, 0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull};
    run<Bn254>(bn_g);
    run<Bls12_381>(bls_g);
    std::printf("all ok\n");
    return 0;
}
