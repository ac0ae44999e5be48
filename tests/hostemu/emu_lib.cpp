// TEST INFRASTRUCTURE ONLY: runs the *device* code of smrt_amd/csrc/dort_device.hpp on the CPU under the fiber
// emulator, one workgroup at a time, behind a tiny C entry point.  Lets the CPU test-suite exercise the kernel
// logic (and check it is barrier-order independent) in a container without a GPU.  Never loaded by smrt_amd.
#define SMRT_HOST_EMU 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../smrt_amd/csrc/dort_active.hpp"
#include "../../smrt_amd/csrc/dort_host_common.hpp"
#include "../../smrt_amd/csrc/dort_phase_kernel.hpp"
#include "../../smrt_amd/csrc/dort_finish_reg.hpp"
#include "../../smrt_amd/csrc/dort_finish_strip.hpp"
#include "../../smrt_amd/csrc/dort_eig_kernel.hpp"
#include "../../smrt_amd/csrc/dort_rayleigh_kernel.hpp"

using namespace smrt;

// 1: three-kernel pipeline with the two-slot finish kernel (default, like the library), 2: three-kernel pipeline with
// the LDS-resident finish kernel, 3: ... with the register-resident finish kernel (passive, N <= 64), 0: fused kernel
extern "C" { int smrt_emu_pipeline = 1; }
// 1: the symmetric eigensolver (dort_eig_kernel.hpp) in place of the Jacobi kernel on the N <= 64 pipelines, like the library's default
extern "C" { int smrt_emu_eig = 1; }
// 1: layers with a Rayleigh phase matrix through the closed-form kernel where a strip finish kernel follows (the library's default)
extern "C" { int smrt_emu_rayleigh = 1; }

// tridiag -> chase -> vectors on one staging item, as k_eig.hip launches them (one wavefront, one lane, one wavefront)
template <int NP>
static long run_eig_np(const DevStage& st, long long it, int order) {
    long nb = 0;
    constexpr int NG = ((NP + 15) / 16) * 16;
    nb += emu::run_block(64, order, [&]() { eig_gram_item<NG>(st, it); });
    nb += emu::run_block(64, order, [&]() { eig_tridiag_item<NP>(st, it); });
    if (st.n[it] <= 0) return nb;
    std::vector<double> cl((size_t)2 * kEigChaseLanes * st.vec_stride, NAN);
    nb += emu::run_block(64, order, [&]() { if (emu::tid() == 5) eig_chase_lane(st, it, cl.data() + 5, cl.data() + kEigChaseLanes * st.vec_stride + 5); });
    if (st.n[it] <= 0) return nb;
    std::vector<double> ring((size_t)2 * kEigRingSlots, NAN);
    nb += emu::run_block(64, order, [&]() { eig_vectors_item<NP>(st, it, ring.data()); });
    return nb;
}
static long run_eig_item(const DevStage& st, long long it, int order) {
    const int rows = st.n[it];
    if (rows <= 0) return 0;
    switch ((rows + 7) / 8) {
        case 1: return run_eig_np<8>(st, it, order);
        case 2: return run_eig_np<16>(st, it, order);
        case 3: return run_eig_np<24>(st, it, order);
        case 4: return run_eig_np<32>(st, it, order);
        case 5: return run_eig_np<40>(st, it, order);
        case 6: return run_eig_np<48>(st, it, order);
        case 7: return run_eig_np<56>(st, it, order);
        default: return run_eig_np<64>(st, it, order);
    }
}

template <int NT, int CH>
static long run_pairs(DevBatch& d, int order, size_t lds_doubles, size_t mat_doubles) {
    long nb = 0;
    std::vector<double> lds(lds_doubles), mat(mat_doubles);
    for (long long p = 0; p < d.pair_count; ++p) {
        for (auto& x : lds) x = NAN;  // uninitialised LDS / workspace must never be consumed
        for (auto& x : mat) x = NAN;
        double* gm = mat_doubles ? mat.data() : nullptr;
        nb += emu::run_block(NT, order, [&]() { dort_pair_passive<NT, CH>(d, p, lds.data(), gm); });
    }
    return nb;
}

template <int NT, int CH>
static long run_active(DevBatch& d, int order, size_t lds_doubles, size_t mat_doubles) {
    long nb = 0;
    std::vector<double> lds(lds_doubles), mat(mat_doubles);
    for (long long p = 0; p < d.pair_count; ++p) {
        for (auto& x : lds) x = NAN;
        for (auto& x : mat) x = NAN;
        double* gm = mat_doubles ? mat.data() : nullptr;
        nb += emu::run_block(NT, order, [&]() { dort_pair_active<NT, CH>(d, p, lds.data(), gm); });
    }
    return nb;
}

// The three-kernel pipelines under emulation, like launch_pipeline of the library: prep for every pair, Jacobi for every
// staged item, finish for every pair -- in up to four rounds over successive layer ranges with the pruning marks in
// between when prune_deep_snowpack is set (the layers below a cut are never staged).
struct Staging {
    std::vector<double> L, B, d, sigma, inv, ws, eig_e, eig_rot;
    std::vector<int> n;
    DevStage st;
    Staging(size_t items, const LdsPlan& plan) : L(items * (size_t)plan.NMAX * plan.LD, NAN), B(L.size(), NAN), d(items * plan.NMAX, NAN),
                                                 sigma(items * plan.NMAX, NAN), inv(items * (plan.NMAX <= 64 ? 1024 : 2048), NAN), ws(plan.NMAX <= 128 ? items * 16384 : 0, NAN), n(items, -1) {
        st = DevStage{L.data(), B.data(), d.data(), sigma.data(), n.data(), (long long)plan.NMAX * plan.LD, plan.NMAX, plan.NMAX <= 128 ? inv.data() : nullptr, ws.data(),
                      plan.NMAX <= 64 ? 1024 : 2048};
        if (plan.NMAX <= 64) {   // the symmetric eigensolver's own staging
            eig_e.assign(items * 2 * plan.NMAX, NAN); eig_rot.assign(items * (size_t)eig_rot_doubles(plan.NMAX), NAN);
            st.eig_e = eig_e.data(); st.eig_rot = eig_rot.data(); st.rot_stride = eig_rot_doubles(plan.NMAX);
        }
    }
};

template <class Prep, class Jac, class Fin>
static long run_rounds(DevBatch& d, int order, int nmodes, Staging& sg, Prep prep, Jac jac, Fin fin) {
    long nb = 0;
    const int rounds = (d.prune_tau > 0.0 && !d.coherent) ? (d.Lmax < 4 ? d.Lmax : 4) : 1;
    if (d.coherent) for (auto& x : sg.n) x = 0;
    std::vector<int> done((size_t)d.pair_count, 0);
    if (rounds > 1) { for (auto& x : sg.n) x = 0; d.pair_done = done.data(); }
    for (int r = 0; r < rounds; ++r) {
        d.layer_lo = (int)((long long)d.Lmax * r / rounds);
        d.layer_hi = (int)((long long)d.Lmax * (r + 1) / rounds);
        for (long long p = 0; p < d.pair_count; ++p) nb += prep(p);
        const long long blocks = d.pair_count * nmodes * (d.layer_hi - d.layer_lo);
        for (long long blk = 0; blk < blocks; ++blk) {
            const long long item = jacobi_item_of_block(d, blk);
            if (sg.n[item] <= 0) continue;   // nothing staged (beyond the snowpack, rejected by prep, or below a cut)
            if (stage_direct(sg.n[item])) {   // a layer with a Rayleigh phase matrix: the closed-form kernel (k_rayleigh.hip)
                std::vector<double> rl((size_t)rayleigh_lds_doubles(), NAN);
                nb += emu::run_block(128, order, [&]() { dort_rayleigh_item<128>(d, sg.st, item, rl.data()); });
                continue;
            }
            nb += jac(item);
        }
        if (r + 1 < rounds)
            for (long long p = 0; p < d.pair_count; ++p)
                nb += emu::run_block(64, order, [&]() { prune_mark_pair(d, sg.st, p, done.data()); });
    }
    d.layer_lo = 0; d.layer_hi = d.Lmax; d.pair_done = nullptr;
    for (long long p = 0; p < d.pair_count; ++p) nb += fin(p);
    return nb;
}

// 64 < N <= 128: prep / finish on a global workspace
template <int NT, bool ACTIVE>
static long run_split_gmem(DevBatch& d, int order, const LdsPlan& plan) {
    const int nmodes = ACTIVE ? d.m_max + 1 : 1;
    Staging sg((size_t)d.pair_count * d.Lmax * nmodes, plan);
    std::vector<double> lds(2 * plan.total + finish_strip_lds_doubles(d.n_max_stream, d.Lmax)), ws((size_t)plan.mat_doubles + plan.scratch_doubles);   // generous: the kernels lay out LDS differently
    const JacobiPlan jp = make_jacobi_plan(d.n_max_stream, ACTIVE ? 3 : 2);
    // the strip finish kernel (one workgroup of eight wavefronts per pair) where the library uses it
    const bool strip = !ACTIVE && smrt_emu_pipeline == 3 && !d.host_itf_slot && !d.coherent && d.sub_kind != SUB_HOST;
    d.rayleigh_direct = (strip && smrt_emu_rayleigh) ? 1 : 0;   // like smrt_dort_upload: where the strip finish kernel reads the staging area
    std::vector<double> jl(jp.total);
    auto fresh = [&]() { for (auto& x : lds) x = NAN; for (auto& x : ws) x = NAN; };
    // (passive: the LDS-resident prep kernel with eight wavefronts where its two packed triangles fit, like the library)
    const size_t wide_lds = (size_t)make_plan(d.n_max_stream, 2, d.Lmax, d.n_theta, 9, 1, 0, 3).total;
    const bool wide = !ACTIVE && wide_lds * sizeof(double) <= 160 * 1024;
    if (wide && lds.size() < wide_lds) lds.resize(wide_lds);
    return run_rounds(d, order, nmodes, sg,
        [&](long long p) { fresh();
                           if (wide) return emu::run_block(512, order, [&]() { dort_pair_passive<512, 1, 1>(d, p, lds.data(), nullptr, &sg.st); });
                           return ACTIVE ? emu::run_block(NT, order, [&]() { dort_pair_active<NT, 2, 1>(d, p, lds.data(), ws.data(), &sg.st); })
                                         : emu::run_block(NT, order, [&]() { dort_pair_passive<NT, 2, 1>(d, p, lds.data(), ws.data(), &sg.st); }); },
        // (the product launches the Jacobi kernel of this pipeline with eight wavefronts, eight lanes per column pair; sixteen and
        // sixteen behind an experiment switch: k_jacobi.hip)
        [&](long long it) { const size_t l16 = (size_t)make_jacobi_plan(d.n_max_stream, ACTIVE ? 3 : 2, 0, 16).total;
                            if (getenv("SMRT_EMU_JACOBI_1024") && l16 * sizeof(double) <= 160 * 1024) {   // (the library's experiment switch SMRT_DORT_JACOBI_1024)
                                std::vector<double> j16(l16, NAN);
                                return emu::run_block(1024, order, [&]() { dort_jacobi_item16<1024>(d, sg.st, it, j16.data()); });
                            }
                            for (auto& x : jl) x = NAN; return emu::run_block(512, order, [&]() { dort_jacobi_item<512>(d, sg.st, it, jl.data()); }); },
        // (and the finish kernel with 512 threads too: k_gmem_split.hip)
        [&](long long p) { fresh();
                           if (strip) return emu::run_block(512, order, [&]() { dort_pair_passive_strip(d, p, lds.data(), sg.st); });
                           return ACTIVE ? emu::run_block(512, order, [&]() { dort_pair_active<512, 2, 2>(d, p, lds.data(), ws.data(), &sg.st); })
                                         : emu::run_block(512, order, [&]() { dort_pair_passive<512, 2, 2>(d, p, lds.data(), ws.data(), &sg.st); }); });
}

// 128 < N <= 384: prep / finish with CH row chunks on the global workspace, the blocked Jacobi kernel in between
template <int CH, bool ACTIVE>
static long run_split_big(DevBatch& d, int order, const LdsPlan& plan) {
    const int nmodes = ACTIVE ? d.m_max + 1 : 1;
    Staging sg((size_t)d.pair_count * d.Lmax * nmodes, plan);
    // (the finish kernels add their operand staging buffers: make_plan jac_in_lds = 3)
    std::vector<double> lds(2 * plan.total + 2 * 64 * ((plan.NMAX + 3) / 4)), ws((size_t)plan.mat_doubles + plan.scratch_doubles);
    const JacobiBigPlan jp = make_jacobi_big_plan(d.n_max_stream, ACTIVE ? 3 : 2);
    std::vector<double> jl(jp.total);
    auto fresh = [&]() { for (auto& x : lds) x = NAN; for (auto& x : ws) x = NAN; };
    return run_rounds(d, order, nmodes, sg,
        [&](long long p) { fresh(); return ACTIVE ? emu::run_block(512, order, [&]() { dort_pair_active<512, CH, 1>(d, p, lds.data(), ws.data(), &sg.st); })
                                                  : emu::run_block(512, order, [&]() { dort_pair_passive<512, CH, 1>(d, p, lds.data(), ws.data(), &sg.st); }); },   // (512 threads like the product)
        [&](long long it) { for (auto& x : jl) x = NAN;
                            return emu::run_block(SMRT_JACOBI_BIG_NT, order, [&]() { dort_jacobi_big_item<SMRT_JACOBI_BIG_NT>(d, sg.st, it, jl.data()); }); },
        // (finish kernels of this pipeline: 512 threads in the product, k_gmem_split_big.hip)
        [&](long long p) { fresh(); return ACTIVE ? emu::run_block(512, order, [&]() { dort_pair_active<512, CH, 2>(d, p, lds.data(), ws.data(), &sg.st); })
                                                  : emu::run_block(512, order, [&]() { dort_pair_passive<512, CH, 2>(d, p, lds.data(), ws.data(), &sg.st); }); });
}

// The Jacobi stage of the N <= 64 pipelines as the library launches it (k_jacobi.hip): one pass per size class of
// jacobi_classes over the item, each with the LDS layout of its class; an item outside the class leaves at once.
static long run_jacobi_classes(DevBatch& d, const DevStage& st, long long it, int P, int order, std::vector<double>& jl) {
    if (smrt_emu_eig && st.eig_rot && P == 2) return run_eig_item(st, it, order);   // passive mode only, like the library
    JacobiClass cls[4];
    const int n = jacobi_classes(d.n_max_stream * P, cls);
    long nb = 0;
    for (int i = 0; i < n; ++i) {
        jl.assign((size_t)make_jacobi_plan(d.n_max_stream, P, cls[i].hi).total, NAN);
        if (cls[i].hi == 32) nb += emu::run_block(jacobi_class_nt(32), order, [&]() { dort_jacobi_item<jacobi_class_nt(32), 0, 32>(d, st, it, jl.data()); });
        else if (cls[i].hi == 48) nb += emu::run_block(jacobi_class_nt(48), order, [&]() { dort_jacobi_item<jacobi_class_nt(48), 32, 48>(d, st, it, jl.data()); });
        else if (cls[i].hi == 56) nb += emu::run_block(jacobi_class_nt(56), order, [&]() { dort_jacobi_item<jacobi_class_nt(56), 48, 56>(d, st, it, jl.data()); });
        else nb += emu::run_block(jacobi_class_nt(64), order, [&]() { dort_jacobi_item<jacobi_class_nt(64), 56, 64>(d, st, it, jl.data()); });
    }
    return nb;
}

// active mode through the three-kernel pipeline: staging items are (pair, azimuth mode, layer)
template <int NT>
static long run_split_active(DevBatch& d, int order, size_t lds_doubles, const LdsPlan& plan) {
    const int nmodes = d.m_max + 1;
    Staging sg((size_t)d.pair_count * d.Lmax * nmodes, plan);
    std::vector<double> lds(lds_doubles);
    const JacobiPlan jp = make_jacobi_plan(d.n_max_stream, 3);
    std::vector<double> jl(jp.total);
    return run_rounds(d, order, nmodes, sg,
        [&](long long p) { for (auto& x : lds) x = NAN; return emu::run_block(NT, order, [&]() { dort_pair_active<NT, 1, 1>(d, p, lds.data(), nullptr, &sg.st); }); },
        [&](long long it) { return run_jacobi_classes(d, sg.st, it, 3, order, jl); },
        [&](long long p) { for (auto& x : lds) x = NAN; return emu::run_block(NT, order, [&]() { dort_pair_active<NT, 1, 3>(d, p, lds.data(), nullptr, &sg.st); }); });
}

// passive, N <= 64: the LDS-resident pipeline (two-slot or four-slot finish)
template <int NT>
static long run_split(DevBatch& d, int order, size_t lds_doubles, const LdsPlan& plan) {
    Staging sg((size_t)d.pair_count * d.Lmax, plan);
    std::vector<double> lds(lds_doubles + finish_reg_lds_doubles(d.n_max_stream, d.Lmax) + finish_strip_lds_doubles(d.n_max_stream, d.Lmax, 4));
    const JacobiPlan jp = make_jacobi_plan(d.n_max_stream, 2);
    std::vector<double> jl(jp.total);
    d.rayleigh_direct = (smrt_emu_pipeline == 6 && smrt_emu_rayleigh && (d.layer_kind != nullptr || em_has_rayleigh_phase(d.emmodel)) && !d.host_itf_slot && !d.coherent && d.sub_kind != SUB_HOST) ? 1 : 0;
    return run_rounds(d, order, 1, sg,
        [&](long long p) { for (auto& x : lds) x = NAN; return emu::run_block(NT, order, [&]() { dort_pair_passive<NT, 1, 1>(d, p, lds.data(), nullptr, &sg.st); }); },
        [&](long long it) { return run_jacobi_classes(d, sg.st, it, 2, order, jl); },
        [&](long long p) { for (auto& x : lds) x = NAN;
                           if (smrt_emu_pipeline == 6 && !d.host_itf_slot && !d.coherent && d.sub_kind != SUB_HOST)   // the strip finish kernel on four wavefronts
                               return d.rayleigh_direct ? emu::run_block(256, order, [&]() { dort_pair_passive_strip4<true>(d, p, lds.data(), sg.st); })
                                                       : emu::run_block(256, order, [&]() { dort_pair_passive_strip4<false>(d, p, lds.data(), sg.st); });   // (the two instances of k_finish_strip.hip)
                           if (smrt_emu_pipeline == 3 && !d.host_itf_slot && !d.coherent && d.sub_kind != SUB_HOST)   // the register-resident finish kernel: one wavefront per pair (where the library uses it)
                               return emu::run_block(64, order, [&]() { dort_pair_passive_reg(d, p, lds.data(), sg.st); });
                           return smrt_emu_pipeline == 2 ? emu::run_block(NT, order, [&]() { dort_pair_passive<NT, 1, 2>(d, p, lds.data(), nullptr, &sg.st); })
                                                         : emu::run_block(NT, order, [&]() { dort_pair_passive<NT, 1, 3>(d, p, lds.data(), nullptr, &sg.st); }); });
}


extern "C" int smrt_emu_run(const smrt_batch* b, long long pair_begin, long long pair_count, int nt, int order,
                            double* out, int32_t* status, double* layer_out, double* stream_out, double* n3_out,
                            long* n_barriers) {
    const char* why = smrt_host::validate(b);
    if (why) { fprintf(stderr, "smrt_emu_run: %s\n", why); return -1; }
    const bool active = (b->mode == SMRT_MODE_ACTIVE);
    const int P = active ? 3 : 2;
    // (pipeline 5, tests only: the global-workspace pipeline with the strip finish kernel whatever the size, so that the
    // small fixtures exercise it; it then behaves like pipeline 3)
    const bool force_gmem = (smrt_emu_pipeline == 5);
    struct Restore { int v; ~Restore() { smrt_emu_pipeline = v; } } restore_pipeline{smrt_emu_pipeline};
    if (force_gmem) smrt_emu_pipeline = 3;
    const bool gmem = b->n_max_stream * P > 64 || force_gmem;
    auto plan_with = [&](int jac) {
        return active ? make_plan(b->n_max_stream, 3, b->n_layers_max, b->n_theta, azimuth_samples(b->m_max) / 2 + 1,
                                  gmem ? 0 : 1, active_doubles(b->n_max_stream, b->n_layers_max, b->n_theta), 0, jac)
                      : make_plan(b->n_max_stream, 2, b->n_layers_max, b->n_theta, 9, gmem ? 0 : 1, 0, 0, jac);
    };
    LdsPlan plan = plan_with(gmem ? 1 : 0);
    int jac_in_lds = gmem ? 1 : 0;
    if (gmem && (size_t)plan.total * sizeof(double) > 160 * 1024) { plan = plan_with(0); jac_in_lds = 0; }   // like smrt_dort_upload
    if (plan.NMAX > 384) return -2;
    const size_t matd = gmem ? (size_t)plan.mat_doubles + plan.scratch_doubles : 0;
    std::vector<double> gl(b->n_max_stream);
    smrt_host::gauss_legendre_positive(b->n_max_stream, gl.data(), nullptr);
    DevBatch d{};
    d.S = b->n_snowpacks; d.Lmax = b->n_layers_max; d.F = b->n_frequencies; d.n_theta = b->n_theta;
    d.emmodel = b->emmodel; d.micro = b->microstructure; d.mode = b->mode; d.n_max_stream = b->n_max_stream;
    d.m_max = b->m_max; d.normalization = b->phase_normalization; d.rayleigh_jeans = b->rayleigh_jeans;
    d.want_layer_out = layer_out ? 1 : 0; d.want_stream_out = stream_out ? 1 : 0;
    d.jac_in_lds = jac_in_lds;
    d.pair_begin = pair_begin; d.pair_count = pair_count;
    d.n_layers = b->n_layers; d.thickness = b->thickness; d.frac_volume = b->frac_volume;
    d.temperature = b->temperature; d.p1 = b->micro_p1; d.p2 = b->micro_p2 ? b->micro_p2 : b->micro_p1;
    d.frequency = b->frequency; d.theta = b->theta; d.gl_mu = gl.data(); d.phi = b->phi;
    d.layer_kind = b->layer_kind;
    d.liquid_water = b->liquid_water;
    d.host_layer = b->host_layer; d.host_coeff = b->host_iba_coeff; d.host_streams = b->host_streams; d.host_phase = b->host_phase;
    d.host_modes = active ? b->m_max + 1 : 1; d.host_ne = b->n_max_stream * (active ? 3 : 2);
    d.coherent = b->process_coherent_layers ? 1 : 0;
    d.host_substrate = b->host_substrate; d.host_substrate_coh = b->host_substrate_coh;
    d.host_itf_slot = b->host_interface_slot; d.host_itf = b->host_interface; d.host_itf_coh = b->host_interface_coh;
    d.host_itf_slots = b->host_interface_slot ? b->host_interface_slots : 0;
    d.sub_kind = b->substrate_kind; d.sub_p1 = b->substrate_p1; d.sub_p2 = b->substrate_p2; d.sub_T = b->substrate_temperature;
    const bool has_atm = b->atm_tb_down != nullptr && b->mode == SMRT_MODE_PASSIVE;
    d.atm_down = has_atm ? b->atm_tb_down : nullptr; d.atm_up = has_atm ? b->atm_tb_up : nullptr;
    d.atm_trans = has_atm ? b->atm_transmittance : nullptr;
    d.prune_tau = (b->prune_optical_depth > 0.0) ? b->prune_optical_depth : 0.0;
    d.layer_lo = 0; d.layer_hi = b->n_layers_max; d.pair_done = nullptr;
    const bool reg_thr = !active && (smrt_emu_pipeline == 3 || smrt_emu_pipeline == 6) && b->n_max_stream * 2 <= 128 && !b->host_interface_slot &&
                         !b->process_coherent_layers && b->substrate_kind != SUB_HOST;   // where the register-resident finish kernel runs
    d.jacobi_skip2 = active ? 1e-30 : (reg_thr ? SMRT_JACOBI_REG_SKIP_COS2 : SMRT_JACOBI_PASSIVE_SKIP_COS2);   // like smrt_dort_upload
    d.jacobi_exit2 = active ? 1e-22 : (reg_thr ? SMRT_JACOBI_REG_EXIT_COS2 : SMRT_JACOBI_PASSIVE_EXIT_COS2);
    d.out = out; d.status = status; d.layer_out = layer_out; d.stream_out = stream_out; d.n3_out = n3_out; d.stage_out = nullptr;
    long nb;
    if (gmem && plan.NMAX > 128 && smrt_emu_pipeline) {   // the big pipeline (blocked Jacobi kernel)
        if (nt != 256) return -3;
        d.jac_in_lds = 0;
        const LdsPlan bp = plan_with(0);
        if (active) nb = plan.NMAX <= 256 ? run_split_big<4, true>(d, order, bp) : run_split_big<6, true>(d, order, bp);
        else nb = plan.NMAX <= 256 ? run_split_big<4, false>(d, order, bp) : run_split_big<6, false>(d, order, bp);
    } else if (gmem && plan.NMAX > 128) {   // fused global-workspace kernels, four / six 64-row chunks
        if (nt != 256) return -3;
        const int ch = plan.NMAX <= 256 ? 4 : 6;
        if (active) nb = ch == 4 ? run_active<256, 4>(d, order, plan.total, matd) : run_active<256, 6>(d, order, plan.total, matd);
        else nb = ch == 4 ? run_pairs<256, 4>(d, order, plan.total, matd) : run_pairs<256, 6>(d, order, plan.total, matd);
    } else if (gmem && smrt_emu_pipeline && plan.NMAX <= 128 && nt == 256) {
        d.jac_in_lds = 1;   // like smrt_dort_upload: the finish kernel of this pipeline always has its LDS scratch
        nb = active ? run_split_gmem<256, true>(d, order, plan) : run_split_gmem<256, false>(d, order, plan);
    } else if (active) {
        if (gmem) {
            switch (nt) {
                case 64: nb = run_active<64, 2>(d, order, plan.total, matd); break;
                case 256: nb = run_active<256, 2>(d, order, plan.total, matd); break;
                default: return -3;
            }
        } else if (smrt_emu_pipeline) {
            switch (nt) {
                case 64: nb = run_split_active<64>(d, order, plan.total, plan); break;
                case 128: nb = run_split_active<128>(d, order, plan.total, plan); break;
                case 256: nb = run_split_active<256>(d, order, plan.total, plan); break;
                default: return -3;
            }
        } else {
            switch (nt) {
                case 64: nb = run_active<64, 1>(d, order, plan.total, 0); break;
                case 128: nb = run_active<128, 1>(d, order, plan.total, 0); break;
                case 256: nb = run_active<256, 1>(d, order, plan.total, 0); break;
                default: return -3;
            }
        }
    } else if (gmem) {
        switch (nt) {
            case 64: nb = run_pairs<64, 2>(d, order, plan.total, matd); break;
            case 256: nb = run_pairs<256, 2>(d, order, plan.total, matd); break;
            default: return -3;
        }
    } else if (smrt_emu_pipeline) {
        switch (nt) {
            case 64: nb = run_split<64>(d, order, plan.total, plan); break;
            case 128: nb = run_split<128>(d, order, plan.total, plan); break;
            case 256: nb = run_split<256>(d, order, plan.total, plan); break;
            default: return -3;
        }
    } else {
        switch (nt) {
            case 64: nb = run_pairs<64, 1>(d, order, plan.total, 0); break;
            case 128: nb = run_pairs<128, 1>(d, order, plan.total, 0); break;
            case 256: nb = run_pairs<256, 1>(d, order, plan.total, 0); break;
            default: return -3;
        }
    }
    if (n_barriers) *n_barriers = nb;
    return 0;
}

extern "C" int smrt_emu_gauss_legendre(int n, double* mu, double* w) {
    smrt_host::gauss_legendre_positive(n, mu, w);
    return 0;
}

// panels of the Gauss-Jordan solves since the last call: [0] pivots from the diagonal block, [1] full-pivot fallback
extern "C" void smrt_emu_panel_counts(long* out2) {
    out2[0] = smrt::smrt_emu_panels[0]; out2[1] = smrt::smrt_emu_panels[1];
    smrt::smrt_emu_panels[0] = smrt::smrt_emu_panels[1] = 0;
}

// The blocked Jacobi kernel (N > 128) on one matrix: Bm is [N][LD] column-major with LD = (n_max_stream * P + 1) | 1,
// rotated in place; sigma [N] out.  Returns the status the kernel left in the staging slot (N = ok, -1 = no convergence).
extern "C" int smrt_emu_jacobi_big(int n_max_stream, int P, int N, double* Bm, double* sigma, int order, double skip2, double exit2) {
    DevBatch d{};
    d.S = 1; d.Lmax = 1; d.F = 1; d.mode = (P == 3) ? 1 : 0; d.m_max = 0; d.n_max_stream = n_max_stream;
    int nl = 1, status = 0;
    d.n_layers = &nl; d.status = &status; d.pair_begin = 0; d.pair_count = 1;
    d.jacobi_skip2 = skip2; d.jacobi_exit2 = exit2;
    const JacobiBigPlan jp = make_jacobi_big_plan(n_max_stream, P);
    int n = N;
    DevStage st{nullptr, Bm, nullptr, sigma, &n, (long long)jp.NMAX * jp.LD, jp.NMAX, nullptr};
    std::vector<double> jl(jp.total, NAN);
    emu::run_block(SMRT_JACOBI_BIG_NT, order, [&]() { dort_jacobi_big_item<SMRT_JACOBI_BIG_NT>(d, st, 0, jl.data()); });
    return n;
}

// The symmetric eigensolver on one matrix: Bm is [N][LD] column-major with LD = (NMAX + 1) | 1, replaced by B' = U Sigma;
// sigma [NMAX] out.  Returns N, or the negative status the kernels left in the staging slot.
extern "C" int smrt_emu_eig_item(int NMAX, int N, double* Bm, double* sigma, int order, long long* n_rotations) {
    int n = N;
    std::vector<double> e((size_t)2 * NMAX, NAN), rot((size_t)eig_rot_doubles(NMAX), NAN);
    DevStage st{nullptr, Bm, nullptr, sigma, &n, (long long)NMAX * ((NMAX + 1) | 1), NMAX, nullptr, nullptr, 1024, e.data(), rot.data(), eig_rot_doubles(NMAX)};
    run_eig_item(st, 0, order);
    if (n_rotations) {   // records of the list (identity padding included)
        long long cnt = 0;
        const double* list = rot.data();
        while (n > 0) {
            const int gt = ((const int*)list)[0], gb = ((const int*)list)[1];
            if (gt < 0) break;
            cnt += 4 * (gt - gb + 1);
            list += 2 + 32 * ((gt >> 2) - (gb >> 2) + 1);
        }
        *n_rotations = cnt;
    }
    return n;
}

// The Rayleigh closed-form kernel on one layer: ke, pa, n cosines (descending), 2 n row scalings in; A+ = D V ([N][LD]
// column-major, LD = (NMAX + 1) | 1), sigma [N] and 1 / D^2 [N] out.  Returns N, or the negative status of the layer.
extern "C" int smrt_emu_rayleigh_item(int NMAX, int n, double ke, double pa, const double* mu, const double* u, double* Ap, double* sigma,
                                      double* inv_d2, int order) {
    const int N = 2 * n;
    std::vector<double> slot(2048, NAN), dvec((size_t)NMAX, 1.0);
    slot[0] = ke; slot[1] = pa;
    for (int j = 0; j < n; ++j) slot[2 + j] = mu[j];
    for (int r = 0; r < N; ++r) slot[2 + n + r] = u[r];
    int nst = N + kStageDirect, nl = 1, status = 0;
    DevBatch b{};
    b.S = 1; b.Lmax = 1; b.F = 1; b.n_layers = &nl; b.status = &status; b.pair_count = 1;
    DevStage st{nullptr, Ap, dvec.data(), sigma, &nst, (long long)NMAX * ((NMAX + 1) | 1), NMAX, slot.data(), nullptr, 2048};
    std::vector<double> lds((size_t)rayleigh_lds_doubles(), NAN);
    emu::run_block(128, order, [&]() { dort_rayleigh_item<128>(b, st, 0, lds.data()); });
    if (nst > 0) for (int r = 0; r < N; ++r) inv_d2[r] = slot[r];
    return nst > 0 ? stage_rows(nst) : nst;
}

// ft_even_phase of one layer through the device function (one emulated thread per (scattered, incident) pair)
extern "C" int smrt_emu_ft_even_phase(int em, int ms, double frequency, double fv, double T, double p1, double p2,
                                      const double* mu_s, int n_s, const double* mu_i, int n_i, int m_max, int npol, double* out) {
    int status = 0;
    PhaseRequest q{em, ms, frequency, fv, T, p1, p2, mu_s, n_s, mu_i, n_i, m_max, npol, azimuth_samples(m_max), out, &status};
    for (int is = 0; is < n_s; ++is)
        for (int ii = 0; ii < n_i; ++ii) ft_even_phase_entry(q, is, ii);
    return status;
}
