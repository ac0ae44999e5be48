/*
 * smrt_dort.h -- C ABI of the MI355X-native DORT hot path (libsmrt_dort.so).
 *
 * The reference (smrt-model/smrt) is pure Python and has no FFI; the interface these entry points replace is the
 * reference's plugin surface for this path (all paths relative to /root/reference):
 *
 *   smrt_dort_run()      <->  one call of runner(function, argument_list)        smrt/core/model.py:395-398
 *                             = for every (sensor_f, snowpack): Model.run_single_simulation
 *                                                                                smrt/core/model.py:584-619
 *                             = prepare_emmodels (IBA / DMRT_QCA_ShortRange ctor) smrt/core/model.py:529-582,
 *                               smrt/emmodel/iba.py:85-137, smrt/emmodel/dmrt_qca_shortrange.py:65-112
 *                             + DORT.solve                                       smrt/rtsolver/dort.py:189-261
 *   smrt_batch           <->  the flattened (frequency-major) simulation list    smrt/core/model.py:476-527
 *                             and the DORT constructor options                   smrt/rtsolver/dort.py:148-178
 *   status[]             <->  SMRTError / error_handling="nan"                   smrt/rtsolver/dort.py:327-334
 *   layer_out/stream_out <->  Result.other_data                                  smrt/rtsolver/rtsolver_utils.py:338-342,373-398
 *
 * Plain C types only: the caller owns every host buffer (C-contiguous float64 / int32); the library owns the
 * device memory inside the context.  One context per GPU, one host thread per context, no global state.
 */
#ifndef SMRT_DORT_H
#define SMRT_DORT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct smrt_dort_ctx smrt_dort_ctx;

/* emmodel (smrt/emmodel/iba.py, dmrt_qca_shortrange.py, dmrt_qcacp_shortrange.py, nonscattering.py) */
#define SMRT_EM_IBA 0
#define SMRT_EM_DMRT_QCA_SHORTRANGE 1
#define SMRT_EM_DMRT_QCACP_SHORTRANGE 2
#define SMRT_EM_NONSCATTERING 3
#define SMRT_EM_HOST 4   /* any other emmodel, evaluated by the caller: see smrt_batch.host_layer / host_phase */
/* IBA with dense_snow_correction="auto" on a layer of more than half ice (smrt/emmodel/iba.py:95-96 ->
 * smrt/core/layer.py:186-201): air inclusions in an ice background.  The layer's frac_volume entry is then the volume
 * fraction of the INCLUSIONS, i.e. 1 - (ice fraction), exactly what the reference's inverted layer carries. */
#define SMRT_EM_IBA_INVERTED 5
/* IBA's PHASE FUNCTION on the layer's microstructure model with the layer's SCALARS from the caller: every emmodel whose
 * phase matrix is (coefficient) x (Fourier transform of the autocorrelation function at 2 k0 Re sqrt(eps_eff) sin(Theta/2)) x
 * (Rayleigh geometry) -- smrt/emmodel/iba.py:228-244 -- and that differs from IBA in the numbers only: iba_original.py
 * (absorption), iba_maxwell_garnett.py (mixing formula, field ratio), derived_IBA(<mixing formula>), IBA on layers with
 * another permittivity model, non-spherical inclusions, a background that is not air (saline ice).  host_layer holds
 * (ks, ka, Re eps_eff, Im eps_eff) of the layer as for SMRT_EM_HOST, host_iba_coeff the coefficient (iba.py:139-150);
 * frac_volume and micro_p1 / micro_p2 are the ones of the (possibly inverted) medium the emmodel works on.  The phase
 * matrices are assembled on the device: no host_streams / host_phase. */
#define SMRT_EM_IBA_HOST 6
/* The RAYLEIGH phase matrix, 3 ks / 2 x the geometry of smrt/emmodel/rayleigh.py:52-127, with the layer's scalars from the
 * caller (host_layer = ks, ka, Re eps_eff, Im eps_eff): every emmodel that inherits ft_even_phase from smrt's Rayleigh --
 * rayleigh.py, sft_rayleigh.py, prescribed_kskaeps.py, the dmrt short-range emmodels on layers the device versions
 * refuse.  The microstructure code and parameters of such a layer are not read. */
#define SMRT_EM_RAYLEIGH_HOST 7
/* microstructure (smrt/microstructure_model/exponential.py, sticky_hard_spheres.py) */
#define SMRT_MS_EXPONENTIAL 0
#define SMRT_MS_STICKY_HARD_SPHERES 1   /* micro_p1 = radius, micro_p2 = stickiness (> 0), or -t: the parameter t itself
                                        * (unified_sticky_hard_spheres.py:24-27; IBA only -- the dmrt emmodels take a stickiness) */
#define SMRT_MS_INDEPENDENT_SPHERE 2   /* micro_p1 = radius (smrt/microstructure_model/independent_sphere.py:54-72); IBA only */
#define SMRT_MS_TEUBNER_STREY 3        /* micro_p1 = corr_length xi, micro_p2 = Y = (2 pi xi / repeat_distance)^2 (teubner_strey.py:45-55); a
                                        * negative Y gives the two-length form of unified_teubner_strey.py:69-72; IBA only */
/* The exponential model evaluated at the COMPLEX wavenumber of the strong-contrast-expansion emmodels (smrt/emmodel/
 * sce_common.py:222-233: 2 k0 sqrt(eps_eff) sin(Theta / 2) with the complex eps_eff, real part of the transform kept,
 * emmodel/common.py:107-117): micro_p1 = corr_length; eps_eff comes from host_layer.  Only as the microstructure code of
 * SMRT_EM_IBA_HOST layers (layer_kind), passive mode.  Teubner-Strey's expression likewise (4 + its own code, micro_p1 /
 * micro_p2 as for SMRT_MS_TEUBNER_STREY).  The codes of the sphere models are reserved and refused: their complex form
 * (sines of the complex k r) cost the headline pipeline's prep kernel more than it was worth; such layers take the
 * SMRT_EM_HOST route. */
#define SMRT_MS_EXPONENTIAL_COMPLEX_K 4
#define SMRT_MS_STICKY_HARD_SPHERES_COMPLEX_K 5
#define SMRT_MS_INDEPENDENT_SPHERE_COMPLEX_K 6
#define SMRT_MS_TEUBNER_STREY_COMPLEX_K 7
/* sensor mode (smrt/core/sensor.py:330-339) */
#define SMRT_MODE_PASSIVE 0
#define SMRT_MODE_ACTIVE 1
/* DORT phase_normalization (smrt/rtsolver/dort.py:94-103): False / "auto"|True / "forced" */
#define SMRT_NORM_OFF 0
#define SMRT_NORM_ON 1
#define SMRT_NORM_FORCED 2

/* substrate under the last layer */
#define SMRT_SUBSTRATE_NONE 0       /* semi-infinite / transparent (rtsolver_utils.py:548-551,601-603) */
#define SMRT_SUBSTRATE_FLAT 1       /* Fresnel against a given permittivity (substrate/flat.py) */
#define SMRT_SUBSTRATE_REFLECTOR 2  /* prescribed specular reflection, emissivity 1 - R (substrate/reflector.py), passive only */
#define SMRT_SUBSTRATE_HOST 3       /* rough substrate: dense reflection matrices (+ emissivity in passive mode) evaluated by the caller */

/* per-pair status word */
#define SMRT_OK 0
#define SMRT_ERR_EIGEN 1           /* eigen iteration did not converge (dort.py:1068-1085)                */
#define SMRT_ERR_NORMALIZATION 2   /* phase renormalisation beyond 30 % (dort.py:792-801)                 */
#define SMRT_ERR_ALBEDO 3          /* single scattering albedo >= 1: no real eigenvalues (dort.py:941)    */
#define SMRT_ERR_SINGULAR 4        /* singular boundary-condition system                                  */
#define SMRT_ERR_INPUT 5           /* invalid layer input (e.g. T > 273.15 K, ice.py:56-57; < 2 streams)   */
#define SMRT_ERR_COHERENT 6        /* process_coherent_layers: the last layer, or two layers in a row, are coherent (coherent_flat.py:26,34) */

/*
 * A batch = S snowpacks x F frequencies, flattened frequency-major exactly like Model.prepare_simulations
 * (pair p = f * S + s).  Per-layer arrays are row-major [S][n_layers_max], layer 0 at the top.
 */
typedef struct smrt_batch {
    int32_t n_snowpacks;      /* S */
    int32_t n_layers_max;     /* row length of the per-layer arrays */
    int32_t n_frequencies;    /* F */
    int32_t n_theta;          /* number of viewing angles (passive) or incidence angles (active) */
    int32_t emmodel;          /* SMRT_EM_* */
    int32_t microstructure;   /* SMRT_MS_* */
    int32_t mode;             /* SMRT_MODE_* */
    int32_t n_max_stream;     /* DORT n_max_stream (dort.py:150) */
    int32_t m_max;            /* DORT m_max, used in active mode only (dort.py:151,209) */
    int32_t phase_normalization; /* SMRT_NORM_* */
    int32_t rayleigh_jeans;   /* DORT rayleigh_jeans_approximation (dort.py:160) */
    int32_t substrate_kind;   /* SMRT_SUBSTRATE_*: what lies under the last layer (Snowpack.substrate) */
    const int32_t* n_layers;  /* [S] */
    const double* thickness;  /* [S][Lmax] m */
    const double* frac_volume;/* [S][Lmax] ice volume fraction (SnowLayer.compute_frac_volumes, make_medium.py:390-434) */
    const double* temperature;/* [S][Lmax] K */
    const double* micro_p1;   /* [S][Lmax] corr_length (exponential) | radius (sticky_hard_spheres), m */
    const double* micro_p2;   /* [S][Lmax] unused (exponential, independent_sphere) | stickiness or -t (sticky_hard_spheres) |
                               * Y (teubner_strey): see SMRT_MS_* */
    const double* frequency;  /* [F] Hz */
    const double* theta;      /* [n_theta] rad: Sensor.theta (== theta_inc in active/backscatter mode) */
    double phi;               /* active: azimuth (rad), pi for backscatter (sensor.py:179-180) */
    /* substrate (smrt/substrate/flat.py, reflector.py; rtsolver_utils.py:544-605, dort.py:429-441), per pair because
     * permittivity / reflection models may depend on the frequency.  Unused (may be NULL) for SMRT_SUBSTRATE_NONE. */
    const double* substrate_p1;          /* [F][S] flat: Re eps_substrate | reflector: specular reflection, V */
    const double* substrate_p2;          /* [F][S] flat: Im eps_substrate | reflector: specular reflection, H */
    const double* substrate_temperature; /* [S] K; <= 0: the substrate does not emit (temperature=None) */
    /* SimpleIsotropicAtmosphere (smrt/atmosphere/simple_isotropic_atmosphere.py; rtsolver_utils.py:251-260,302-305),
     * passive mode only; all three NULL = no atmosphere */
    const double* atm_tb_down;           /* [F] K */
    const double* atm_tb_up;             /* [F] K */
    const double* atm_transmittance;     /* [F] */
    /* DORT option prune_deep_snowpack (dort.py:117-124,176-178,443-452): optical depth (sum over the layers, from the
     * top, of min|beta_l| * thickness_l) beyond which the deeper layers are left out of the solve; the layer in which
     * the threshold is passed keeps its bottom reflection and receives nothing from below.  <= 0 (or NaN): off.
     * Needs a three-kernel pipeline (the eigenvalues of all the layers are known before the boundary recursion starts;
     * every size up to the limit of 384 streams x polarisations has one): smrt_dort_upload fails only after
     * smrt_dort_set_pipeline(ctx, 0).  The prep and Jacobi kernels then run in up to four rounds over successive layer
     * ranges, so the layers below a cut are never diagonalised. */
    double prune_optical_depth;
    /* Heterogeneous snowpacks (a list / dict of emmodels in make_model, per-layer microstructure models in make_snowpack;
     * smrt/core/model.py:529-582): [S][Lmax] emmodel + 16 * microstructure of every layer (SMRT_EM_* + 16 * SMRT_MS_*).
     * NULL: every layer uses `emmodel` / `microstructure` above. */
    const int32_t* layer_kind;
    /* Electromagnetic models evaluated by the caller -- any object with the reference's emmodel protocol
     * (effective_permittivity, ks, ka, ft_even_phase; smrt/rtsolver/dort.py:189,231-247,714-762).  Layers whose kind is
     * SMRT_EM_HOST take their scalars and the azimuth modes of their phase matrix from these arrays instead of
     * evaluating an emmodel on the device; everything else (streams, interfaces, diagonalisation, boundary system) is
     * the same device path.  Indexed by the pair p = f * S + s.  All NULL when no layer is of that kind.
     *   host_layer   [F * S][Lmax][4]: ks, ka (1/m, isotropic), Re and Im of the effective permittivity
     *   host_streams [F * S][Lmax]: number of streams of the layer as the caller computed it (streams.py:136-223);
     *                the device checks it against its own count (SMRT_ERR_INPUT on a mismatch)
     *   host_phase   [F * S][Lmax][modes][2][NE * NE], modes = 1 (passive) or m_max + 1 (active), NE = n_max_stream *
     *                polarisations (2 passive, 3 active): ft_even_phase(mu, +mu') and ft_even_phase(mu, -mu') of
     *                mode m on the layer's own stream cosines, compressed like smrt/core/lib.py:336-347 (row =
     *                stream_s * polarisations + pol_s, column = stream_i * polarisations + pol_i), row-major with
     *                leading dimension NE; only the rows / columns below n_l * polarisations are read.  The matrix
     *                must obey reciprocity (symmetric P(mu,+mu'), P(mu,-mu') up to the sign / factor 2 conventions
     *                of the U polarisation, emmodel/common.py:40-50): the device reads its lower triangle. */
    const double* host_layer;
    const int32_t* host_streams;
    const double* host_phase;
    /* DORT option process_coherent_layers (dort.py:110,156,203; rtsolver_utils.py:349-365; interface/coherent_flat.py):
     * non-zero = per pair, every layer with k0 Re(sqrt(eps_eff)) thickness < 3 pi / 4 at the pair's frequency is taken
     * out of the snowpack and becomes a coherent (Fabry-Perot) interface on top of the layer below it; layer_out then
     * holds the remaining layers, top first, and zeros after them, with 1024 x (index of the layer in the input) added to
     * the stream count of column 4 so that the caller can tell which layers were kept.  With SMRT_EM_HOST layers the
     * host_* arrays stay indexed by the layer's position in the INPUT; host_streams / host_phase of a layer that stays
     * must then be sampled on the streams of the REDUCED snowpack (the caller applies the same criterion to its own
     * permittivities: the most refringent layer is searched among the layers that stay), entries of layers that leave are
     * not read apart from the permittivity. */
    int32_t process_coherent_layers;
    /* SMRT_SUBSTRATE_HOST: a rough substrate (smrt/substrate/geometrical_optics.py, iem_fung92*.py, ...).  The reflection
     * matrix of the bottom boundary is no longer diagonal; the caller evaluates it with the reference's own substrate
     * classes exactly as compute_interface_properties does (rtsolver_utils.py:567-597,690-707: specular_reflection_matrix
     * on the diagonal + 2 pi (mode 0) | pi (mode >= 1) x ft_even_diffuse_reflection_matrix normalised by mu and the
     * stream weights) on the streams of the LAST layer, and the device starts its bottom-up recursion from it.  Indexed
     * by the pair p = f * S + s; modes = m_max + 1 in active mode, 1 in passive mode; NE = 3 * n_max_stream:
     *   host_substrate     [F * S][modes][NE * NE]: reflection_bottom(last layer, mode m), compressed (row = scattered
     *                      stream * P + polarisation, column = incident; P = 2 for mode 0, 3 above), row-major with
     *                      leading dimension NE; rows / columns below n_last * P are read
     *   host_substrate_coh [F * S][modes][NE]: active mode: the diagonal of its specular part (coherent-only pass, mode
     *                      0 is used); passive mode: the emissivity diagonal (substrate.emissivity_matrix,
     *                      rtsolver_utils.py:533-536), multiplied by B(substrate_temperature[s]) on the device
     * (The reference runs its purely diffuse substrates -- geometrical_optics -- in active mode only: in passive mode it
     * raises, dort.py:437; iem_fung92*, geometrical_optics_backscatter run in both.)  Under process_coherent_layers the
     * caller samples them on the streams of the REDUCED snowpack (the layers with k0 Re(n) d >= 3 pi / 4). */
    const double* host_substrate;
    const double* host_substrate_coh;
    /* SMRT_INTERFACE_HOST: rough interfaces at the surface or between layers (smrt/interface/iem_fung92.py,
     * geometrical_optics.py, ...; smrt/rtsolver/rtsolver_utils.py:473-642).  Their reflection / transmission matrices are
     * dense in the streams; the caller evaluates them with the reference's own interface classes exactly as
     * compute_interface_properties combines them (specular / coherent part on the diagonal + 2 pi (mode 0) | pi (mode >= 1)
     * x the diffuse mode normalised by mu and the stream weights) and the device composes each of them with the
     * reflection matrix of everything below (one N x N solve and two products per rough interface and mode).
     *   host_interface_slot [F * S][n_layers_max] int32: -1 = the interface ON TOP of this layer is Flat (Fresnel on the
     *                       device); k >= 0 = it is rough and slot k of host_interface holds it (layer 0: the surface);
     *   host_interface      [F * S][host_interface_slots][modes][4][NE * NE], modes = m_max + 1 in active mode, 1 in passive
     *                       mode, NE = 3 * n_max_stream, row-major with leading dimension NE, compressed order (stream * P +
     *                       polarisation; P = 2 for mode 0, 3 above), zero outside the streams that exist:
     *                         [0] Rtop  reflection_top(layer):       rows and columns = streams of the layer
     *                         [1] Ttop  transmission_top(layer):     rows = streams of the medium above, columns = of the layer
     *                         [2] Rbot  reflection_bottom(above):    rows and columns = streams of the medium above
     *                         [3] Tbot  transmission_bottom(above):  rows = streams of the layer, columns = of the medium above
     *                       (a purely specular transmission is diagonal and is cut to the common streams, like the
     *                       reference does, dort.py:372-376,409-414);
     *   host_interface_coh  [F * S][host_interface_slots][4][NE]: the diagonals of the specular-only versions of the same
     *                       four for mode 0 (index 2 * stream + polarisation): the coherent pass of active mode.
     * NULL host_interface_slot: every interface is Flat.  Under prune_deep_snowpack a rough interface right below the last
     * kept layer closes the recursion with its Rbot (what the reference's truncated system keeps, dort.py:443-452).  Under
     * process_coherent_layers the caller samples the matrices on the streams of the REDUCED snowpack and gives no slot to
     * an interface on top of or right below a collapsed layer (the reference's CoherentFlat takes its place,
     * interface/coherent_flat.py:37-45): the slot index stays the layer's index in the INPUT arrays. */
    const int32_t* host_interface_slot;
    const double* host_interface;
    const double* host_interface_coh;
    int32_t host_interface_slots;
    /* Wet snow (smrt/inputs/make_medium.py:316-434, smrt/permittivity/wetice.py:12-45): liquid_water [S][n_layers_max] =
     * water volume / (ice + water volume) of every layer, or NULL: dry snow.  A wet layer (> 0) must be at the melting
     * point (273.15 K: the water permittivity, Maetzler & Wegmuller 1987, is not defined below it -- status 5 otherwise);
     * its scatterers are ice spheres coated in water: permittivity by Maxwell Garnett with water as the host
     * (wetice_permittivity_bohren83, the reference's default for snow layers); frac_volume is then the volume fraction of
     * ice + water (SnowLayer.compute_frac_volumes).  Applies to the device emmodels; ignored by SMRT_EM_HOST layers. */
    const double* liquid_water;
    /* layers of kind SMRT_EM_IBA_HOST: [F * S][Lmax] the coefficient of IBA's phase matrix (see SMRT_EM_IBA_HOST), indexed
     * like host_layer by the global pair f * S + s; null when no layer has that kind */
    const double* host_iba_coeff;
} smrt_batch;

/* Sizes of the output rows (doubles per pair). Passive: Tb[pol V,H][theta].  Active: I[pol][pol_inc][theta_inc]
 * with 3 x 3 polarisations V,H,U (layout of Result.data in the reference, rtsolver_utils.py:327-332). */
int32_t smrt_dort_out_stride(const smrt_batch* b);

/* Number of visible HIP devices (0 if none). */
int32_t smrt_dort_device_count(void);

/* Create / destroy a context bound to HIP device `device`.  Returns 0 on success. */
int32_t smrt_dort_create(smrt_dort_ctx** ctx, int32_t device);
void smrt_dort_destroy(smrt_dort_ctx* ctx);
const char* smrt_dort_last_error(const smrt_dort_ctx* ctx);

/*
 * One shot: H2D of the packed batch, kernel, D2H.  pair range [pair_begin, pair_begin + pair_count) of the
 * flattened list (pair_count < 0: all).  out: [pair_count][out_stride]; status: [pair_count];
 * layer_out (optional, may be NULL): [pair_count][Lmax][5] = Re eps_eff, Im eps_eff, ks, ka, n_streams;
 * stream_out (optional): [pair_count][1 + n_max_stream] = n_air, outmu[0..n_air).
 */
int32_t smrt_dort_run(smrt_dort_ctx* ctx, const smrt_batch* batch, int64_t pair_begin, int64_t pair_count,
                      double* out, int32_t* status, double* layer_out, double* stream_out);

/*
 * Sparse selection: only the listed pairs of the flattened S x F list (indices p = f * S + s, any order, repeats
 * allowed); out / status / layer_out / stream_out have n_pairs rows, row i belongs to pairs[i].  This is what a runner
 * uses when the simulations it is given are not the full Cartesian product -- a sequence of sensors zipped with a
 * sequence of snowpacks (smrt/core/model.py:505-515), or a list of (sensor, snowpack) pairs handed over one by one.
 */
int32_t smrt_dort_run_pairs(smrt_dort_ctx* ctx, const smrt_batch* batch, const int64_t* pairs, int64_t n_pairs,
                            double* out, int32_t* status, double* layer_out, double* stream_out);

/*
 * Split form for resident data and timing: upload once, launch many times, download.
 * smrt_dort_launch is asynchronous on the context's stream; smrt_dort_sync waits for it.
 * If out_dev / status_dev are non-NULL they are DEVICE pointers (e.g. torch CUDA tensors used as the send
 * buffers of an RCCL gather) that receive the results instead of the context's own buffers.
 */
int32_t smrt_dort_upload(smrt_dort_ctx* ctx, const smrt_batch* batch, int64_t pair_begin, int64_t pair_count);
int32_t smrt_dort_upload_pairs(smrt_dort_ctx* ctx, const smrt_batch* batch, const int64_t* pairs, int64_t n_pairs);
int32_t smrt_dort_launch(smrt_dort_ctx* ctx, void* out_dev, void* status_dev);
int32_t smrt_dort_sync(smrt_dort_ctx* ctx);
int32_t smrt_dort_download(smrt_dort_ctx* ctx, double* out, int32_t* status, double* layer_out,
                           double* stream_out);

/* HIP-event time (ms) of the most recent smrt_dort_launch on the context's stream (valid after sync),
 * and accumulated over all launches since the last reset (count returned through n_launches). */
double smrt_dort_last_kernel_ms(smrt_dort_ctx* ctx);

/* Per-kernel HIP-event time of the three-kernel pipelines.  enable > 0: every prep / diagonalisation / finish kernel launch
 * of the following smrt_dort_launch calls is bracketed by an event pair on its stream (a few microseconds per launch: for
 * measurement runs, not for the timed region of a benchmark); enable == 0: off; enable < 0: leave it as it is (read only).
 * ms3 != NULL: the intervals of the LAST launch summed per kind -- ms3[0] prep, [1] diagonalisation (the Jacobi launches
 * of all size classes, or the gram / tridiag / chase / vectors kernels of the symmetric eigensolver), [2] finish -- after a
 * stream synchronisation; zeros when that launch was not instrumented or the batch runs on a fused kernel.  (With several
 * concurrent pipeline passes the intervals of different passes overlap: the sums then exceed the wall time of the launch.)
 * Returns the number of intervals of the last launch, -1 on error. */
int32_t smrt_dort_kernel_breakdown(smrt_dort_ctx* ctx, int32_t enable, double* ms3);

/* How the layer eigenproblems are diagonalised on the three-kernel pipelines -- the device counterpart of the reference's
 * DORT option diagonalization_method (smrt/rtsolver/dort.py:614,821-962: eig / schur / half_rank_eig / stamnes88; all of them
 * work on the squared problem, like SMRT_DIAG_SYMMETRIC).  mode: SMRT_DIAG_JACOBI, SMRT_DIAG_SYMMETRIC (wherever it is
 * built: passive mode, streams x polarisations <= 64; other batches keep the Jacobi kernels), or -1: the default (SMRT_DIAG_SYMMETRIC;
 * the environment variable SMRT_DORT_EIG=0 / 1 overrides it for experiments).  Takes effect at the next upload. */
int32_t smrt_dort_set_diagonalisation(smrt_dort_ctx* ctx, int32_t mode);

/* What the uploaded batch runs on, as numbers (tests assert the designed kernel choice with these instead of wall-clock
 * ratios): info[SMRT_INFO_*], at most n entries written; returns SMRT_INFO_COUNT, -1 on error.
 *   PIPELINE      SMRT_PIPELINE_*: the kernels a launch of this batch consists of
 *   CHUNK_PAIRS   pairs per pipeline pass (the staging area holds one chunk), CHUNKS: passes per launch
 *   PRUNE_ROUNDS  layer ranges the prep + Jacobi kernels run over under prune_deep_snowpack (1: all layers at once)
 *   STAGED_ITEMS  (pair, azimuth mode, layer) items the LAST chunk of the last launch diagonalised -- known when the
 *                 staging counts are reset per chunk (prune rounds, process_coherent_layers), else -1; synchronises
 *   BLOCK_THREADS workgroup size of the per-pair kernels, N_MAX: padded matrix order (streams x polarisations)
 *   DIAGONALISATION  SMRT_DIAG_*: what runs between the prep and the finish kernels of the three-kernel pipelines
 *   RAYLEIGH_CLOSED_FORM  1: the layers with a Rayleigh phase matrix (dmrt_qca_shortrange, dmrt_qcacp_shortrange,
 *                 nonscattering, rayleigh-family host emmodels) of this batch skip Cholesky and the iteration: their azimuth
 *                 mode 0 is "diagonal minus rank two" and is diagonalised in closed form (passive batches on the strip finish
 *                 kernels; SMRT_DORT_RAYLEIGH=0 switches it off for experiments) */
enum { SMRT_INFO_PIPELINE = 0, SMRT_INFO_CHUNK_PAIRS, SMRT_INFO_CHUNKS, SMRT_INFO_PRUNE_ROUNDS, SMRT_INFO_STAGED_ITEMS,
       SMRT_INFO_BLOCK_THREADS, SMRT_INFO_N_MAX, SMRT_INFO_DIAGONALISATION, SMRT_INFO_RAYLEIGH_CLOSED_FORM, SMRT_INFO_COUNT };
enum { SMRT_DIAG_JACOBI = 0,      /* one-sided Jacobi on B = L+^T L- (singular values to high relative accuracy) */
       SMRT_DIAG_SYMMETRIC = 1 }; /* Householder tridiagonalisation + implicit QL on S = B B^T (passive, N <= 64: the default) */
enum { SMRT_PIPELINE_FUSED = 0,          /* one kernel per pair, matrices in LDS (N <= 64) */
       SMRT_PIPELINE_LDS_TWO_SLOT = 1,   /* prep + Jacobi + two-slot finish, matrices in LDS */
       SMRT_PIPELINE_LDS_FOUR_SLOT = 2,  /* ... with the four-slot finish kernel (set_pipeline(2)) */
       SMRT_PIPELINE_LDS_REG = 3,        /* ... with the register-resident finish kernel (passive default) */
       SMRT_PIPELINE_FUSED_GMEM = 4,     /* one kernel per pair on a global workspace (N > 64) */
       SMRT_PIPELINE_GMEM = 5,           /* prep + Jacobi + finish on the global workspace, 64 < N <= 128 */
       SMRT_PIPELINE_BIG = 6,            /* ... with the blocked Jacobi kernel, 128 < N <= 384 */
       SMRT_PIPELINE_GMEM_STRIP = 7,     /* LDS-resident prep + Jacobi + the strip finish kernel (eight wavefronts), 64 < N <= 128 (passive default) */
       SMRT_PIPELINE_LDS_STRIP = 8 };    /* prep + Jacobi + the strip finish kernel (four wavefronts), N <= 64 (passive default) */
int32_t smrt_dort_launch_info(smrt_dort_ctx* ctx, int64_t* info, int32_t n);
double smrt_dort_total_kernel_ms(smrt_dort_ctx* ctx, int64_t* n_launches, int32_t reset);

/* Tuning knob: threads per workgroup of the per-pair kernels: 64 (one wavefront) or 256 (default, also 0). */
int32_t smrt_dort_set_block_threads(smrt_dort_ctx* ctx, int32_t threads);

/* Pipeline shape.  1 (default) = three kernels -- prep per pair, Jacobi per (pair, layer[, azimuth mode]), finish per
 * pair -- with the factors staged through HBM/L2, for every size (N = streams x polarisations of the batch maximum):
 *   N <= 64:        matrices in LDS; finish = the register-resident kernel (one wavefront per pair, four per CU) in
 *                   passive mode with Flat interfaces, else the two-slot LDS kernel (two workgroups per CU);
 *   64 < N <= 128:  per-workgroup global workspace, Jacobi kernel on a 128-column LDS matrix;
 *   128 < N <= 384: global workspace, blocked Jacobi kernel (the limit of this build: n_max_stream <= 192 passive,
 *                   <= 128 active; smrt_dort_upload fails beyond).
 * In passive mode with Flat interfaces the finish kernel of 1 is the pivot-free recursion: the STRIP kernel (one workgroup
 * of four wavefronts per pair for N <= 64 while three of them share a CU, of eight for 64 < N <= 128), else the
 * register-resident one (one wavefront per pair, N <= 64).
 * 3 = like 1 with the register-resident finish kernel wherever it is supported (N <= 64) instead of the four-wavefront
 * strip kernel; 5 = like 1, the strip kernels wherever they are supported (also at two workgroups per CU); 4 = like 1,
 * never a pivot-free finish kernel (A/B runs); 2 = like 4 with the four-matrix LDS finish kernel (one workgroup per CU;
 * N <= 64 passive only, otherwise like 0); 0 = everything fused in one kernel, one workgroup per pair (no
 * prune_deep_snowpack).  Call before smrt_dort_upload.  In the environment SMRT_DORT_FINISH_REG=0|1 overrides 3 / 4 for
 * the register-resident kernel, SMRT_DORT_FINISH_STRIP=0|1 and SMRT_DORT_FINISH_STRIP4=0|1 for the two strip kernels. */
int32_t smrt_dort_set_pipeline(smrt_dort_ctx* ctx, int32_t split);

/* LDS bytes one workgroup (= one wavefront) of the register-resident finish kernel takes for a batch with this
 * n_max_stream and n_layers_max: four of them share the 160 KB of a CU while this is <= 40 KB (40 layers at 32 streams),
 * three up to 160 KB / 3 (~140 layers) -- the range in which the default pipeline takes this kernel --, two up to the
 * 64 KB a workgroup may have, where only smrt_dort_set_pipeline(ctx, 3) selects it (it ties with the two-slot kernel). */
int32_t smrt_dort_finish_reg_lds_bytes(int32_t n_max_stream, int32_t n_layers_max);

/* LDS bytes one workgroup of the strip finish kernel takes: wavefronts = 4 (N <= 64; the default pipeline takes it while
 * three workgroups share a CU, i.e. up to 160 KB / 3) or 8 (64 < N <= 128: one workgroup per CU, taken while it fits the
 * 160 KB at all).  Returns -1 for other wavefront counts. */
int32_t smrt_dort_finish_strip_lds_bytes(int32_t n_max_stream, int32_t n_layers_max, int32_t wavefronts);

/* LDS bytes one workgroup of the Jacobi kernel takes in the three-kernel pipelines of up to 128 columns (n_pol = 2
 * passive, 3 active).  The N <= 64 pipelines launch one kernel per SIZE CLASS of items (at most 32 / 33-48 / 49-56 /
 * 57-64 columns), each with the layout of its own largest item: size_class_columns = that bound, or 0 for the layout of
 * the batch maximum (the one launch of the 64 < N <= 128 pipeline).  Returns -1 for shapes outside these pipelines. */
int32_t smrt_dort_jacobi_lds_bytes(int32_t n_max_stream, int32_t n_pol, int32_t size_class_columns);

/* Algorithmic work of the uploaded batch after a launch: sum over pairs, modes and layers of N_l^3
 * (N_l = streams x polarisations in layer l), the quantity SURVEY.md 8(d) prices at 68 flops. */
double smrt_dort_sum_n3(smrt_dort_ctx* ctx);

/*
 * The emmodel protocol's ft_even_phase for ONE layer (smrt/emmodel/common.py:349-399, rayleigh.py:52-127; consumed by
 * the reference's rtsolvers at smrt/rtsolver/dort.py:231-247): azimuthal modes 0..m_max of the phase matrix on the grid
 * mu_s (scattered cosines) x mu_i (incident cosines), any signs.  out: [npol][npol][m_max + 1][n_s][n_i], npol = 2 or 3
 * (V, H[, U]).  The layer is described like a row of smrt_batch: emmodel / microstructure codes, frequency (Hz), ice
 * volume fraction, temperature (K), micro_p1 / micro_p2 (corr_length | radius, stickiness).  What lets smrt_amd's
 * emmodel classes serve an rtsolver other than smrt_amd's own DORT (which assembles these modes inside its kernels).
 */
int32_t smrt_dort_ft_even_phase(smrt_dort_ctx* ctx, int32_t emmodel, int32_t microstructure, double frequency,
                                double frac_volume, double temperature, double micro_p1, double micro_p2,
                                const double* mu_s, int32_t n_s, const double* mu_i, int32_t n_i, int32_t m_max,
                                int32_t npol, double* out);

/* Work estimate of every pair of the uploaded batch BEFORE solving it: sum over its layers (and azimuth modes) of
 * N_l^3 from the stream counts alone (a cheap kernel: layer permittivities and Snell's law only).  cost: [pair_count]
 * host doubles; 0 for a pair with invalid input.  What a caller shards by when it splits a batch over several GPUs
 * (total reflection removes streams, so the work of a pair varies with its density profile). */
int32_t smrt_dort_pair_cost(smrt_dort_ctx* ctx, double* cost);

/* Profiling builds only (-DSMRT_STAGE_TIMING): shader cycles of workgroup thread 0 per kernel stage, summed over
 * the pairs of the last launch (setup, assemble, cholesky, L^T L, jacobi, triangular, R1, LU1, R45, LU2, R78, out).
 * A regular build returns zeros. */
int32_t smrt_dort_stage_cycles(smrt_dort_ctx* ctx, double* out16);

/* Positive Gauss-Legendre nodes of order 2n in descending order (smrt/rtsolver/streams.py:300-313). Host only. */
int32_t smrt_gauss_legendre_positive(int32_t n, double* mu, double* weight);

/*
 * Multi-GPU: the (snowpack x frequency) list is embarrassingly parallel (smrt/core/model.py:395-398 maps one function
 * over it; smrt/runner/joblib_runner.py:45-72 is the reference's fan-out), so every GPU solves its own slice and the
 * ONLY communication of the path is the gather of the result rows to one rank -- done here with RCCL over xGMI,
 * device buffer to device buffer, no PyTorch involved.
 *
 * One context = one rank.  Either one process per GPU: rank 0 calls smrt_dort_comm_unique_id, the caller ships the
 * 128 bytes to the other ranks by any means it has (smrt_amd/runner/distributed.py uses a TCP socket on MASTER_ADDR),
 * every rank calls smrt_dort_comm_init (collective).  Or one process driving several GPUs: smrt_dort_comm_init_all on
 * the list of contexts (ncclCommInitAll), then smrt_dort_gather from one host thread per context.
 *
 * smrt_dort_gather (collective): the rows of the LAST launch of every rank (its own out / status buffers, i.e.
 * smrt_dort_launch(ctx, NULL, NULL)) land on `root` in rank order; counts[r] = number of pairs of rank r (the
 * caller's sharding: counts[own rank] must equal the uploaded pair count).  On the root, out [sum counts][out_stride]
 * and status [sum counts] are HOST buffers (either may be NULL to leave the rows on the device); ignored elsewhere.
 * It is enqueued on the context's stream behind the kernels and returns when the root has its rows.
 * smrt_dort_comm_allreduce_max: element-wise maximum over the ranks of n host doubles, in place (the max-over-ranks
 * of a timing; with n = 0 it is a barrier).
 * smrt_dort_comm_library: which RCCL the gather runs on -- the file the symbols were resolved from (dladdr; on failure
 * the reason) copied into path[capacity], and ncclGetVersion's code (e.g. 22703).  The environment variable
 * SMRT_RCCL_LIB pins the library (path or soname; nothing else is tried then); by default a librccl.so.1 the process
 * has already mapped is kept (it belongs to the HIP runtime in use), else /opt/rocm/lib/librccl.so.1, else the soname.
 */
#define SMRT_COMM_ID_BYTES 128
int32_t smrt_dort_comm_library(char* path, int32_t capacity, int32_t* version);
int32_t smrt_dort_comm_unique_id(char* id);
int32_t smrt_dort_comm_init(smrt_dort_ctx* ctx, int32_t world, int32_t rank, const char* id);
int32_t smrt_dort_comm_init_all(smrt_dort_ctx** ctxs, int32_t n);
int32_t smrt_dort_comm_destroy(smrt_dort_ctx* ctx);
int32_t smrt_dort_gather(smrt_dort_ctx* ctx, int32_t root, const int64_t* counts, double* out, int32_t* status);
int32_t smrt_dort_comm_allreduce_max(smrt_dort_ctx* ctx, double* values, int32_t n);

/* The transfers smrt_dort_gather issues, as data (host arithmetic only -- no GPU, no communicator): on `root` one receive
 * per other rank that has rows, the rows of rank r landing at row offset counts[0] + ... + counts[r - 1] of the gathered
 * buffer (own rows are copied to *own_offset_rows); on any other rank one send of its counts[rank] rows.  Returns the
 * number of transfers (ops beyond `capacity` are counted, not written), -1 on invalid arguments; *total_rows = sum of
 * counts.  What the multi-rank CPU tests check with 2..8 ranks, unequal and empty shards and any root. */
typedef struct smrt_gather_op { int32_t peer; int32_t reserved; int64_t offset_rows; int64_t rows; } smrt_gather_op;
int32_t smrt_dort_gather_plan(int32_t world, int32_t root, int32_t rank, const int64_t* counts, smrt_gather_op* ops,
                              int32_t capacity, int64_t* own_offset_rows, int64_t* total_rows);

/* Self-description of the ABI for foreign-function bindings: out[0] = sizeof(smrt_batch), out[1..] = byte offset of
 * every field of smrt_batch in declaration order.  Returns the number of entries of the full description (fields + 1);
 * at most `capacity` of them are written (out may be NULL to query the count).  A binding checks its own struct
 * declaration against this at load time (tests/test_host_logic.py::test_ctypes_struct_matches_the_library). */
int32_t smrt_dort_abi(int32_t* out, int32_t capacity);

const char* smrt_dort_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SMRT_DORT_H */
