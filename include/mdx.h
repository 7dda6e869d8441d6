/*
 * mdx.h — C ABI of the MI355X-native MD force engine that sits behind Molchanica's `src/md`
 * step loop.
 *
 * Every entry point below replaces one use of the (external, absent) `dynamics` crate that the
 * reference makes; the reference call site is cited as  [ref: file:line]  relative to
 * /root/reference.  The Rust host binds these with a plain `extern "C"` block (see
 * INTEGRATION.md); no library-allocated memory crosses the ABI, all outputs are caller-allocated.
 *
 * Units (from the reference UI, src/ui/popup/ff_params.rs:343-344,392-393,465-467; md_viewer.rs:201-256):
 *   length Å, time ps, mass Da, energy kcal/mol, charge e, angles rad, bond k kcal/mol/Å²,
 *   angle k kcal/mol/rad².  Engine state is f32 (src/md/mod.rs:848-852, dt: f32 at :699).
 *
 * Threading: a handle is NOT thread-safe (the reference holds `&mut MdState`, src/md/mod.rs:748);
 * any one thread at a time may call into it.  The library uses its own non-default HIP stream.
 */
#ifndef MDX_H
#define MDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (ParamError{descrip} analogue, src/md/mod.rs:967) ------------------------- */
#define MDX_OK       0
#define MDX_EPARAM  (-1) /* bad system / config (the reference's ParamError)                      */
#define MDX_EDEVICE (-2) /* no usable GPU / HIP error; host falls back like src/util.rs:1072-1119 */
#define MDX_ENAN    (-3) /* non-finite state detected (blow-up; cf. sol_shrinking_box.rs:776-789) */
#define MDX_EOOM    (-4) /* device or host allocation failed                                     */

/* ---- MdOverrides switches  [ref: src/md/mod.rs:671-682; src/mol_editor/mod.rs:869-875] ------ */
#define MDX_OVR_BONDED_DISABLED           0x1u
#define MDX_OVR_COULOMB_DISABLED          0x2u
#define MDX_OVR_LJ_DISABLED               0x4u
#define MDX_OVR_LONG_RANGE_RECIP_DISABLED 0x8u /* always effectively set: cutoff Coulomb only   */

/* ---- per-atom flags  [ref: AtomDynamics.static_/bonded_only, src/docking/mod.rs:260-262] ---- */
#define MDX_ATOM_STATIC      0x1u /* never integrated (infinite mass)                             */
#define MDX_ATOM_BONDED_ONLY 0x2u /* takes part in bonded terms only (no LJ / Coulomb)            */
#define MDX_ATOM_GHOST       0x4u /* multi-GPU halo copy: position is imposed each step, never
                                     integrated, its force is not reported                       */

/* ---- Coulomb treatment inside the cutoff ----------------------------------------------------- */
#define MDX_COULOMB_SHIFTED   0 /* E = k q q (1/r - 1/rc), F = k q q / r^2          (default)     */
#define MDX_COULOMB_REACTION  1 /* reaction field eps_rf = inf: F = k q q (1/r^2 - r/rc^3)        */
#define MDX_COULOMB_EWALD     2 /* Ewald: real space E = k q q erfc(a r)/r in the pair loop, plus the SPME
                                   reciprocal sum (hipFFT) unless MDX_OVR_LONG_RANGE_RECIP_DISABLED is set
                                   [ref: README.md:240; src/util.rs:1094-1100; src/mol_editor/mod.rs:873]  */

#define MDX_COMBINE_LORENTZ_BERTHELOT 0 /* sigma_ij=(si+sj)/2, eps_ij=sqrt(ei ej)   (default)     */
#define MDX_COMBINE_GEOMETRIC         1 /* sigma_ij=sqrt(si sj), eps_ij=sqrt(ei ej)               */

/* which array mdx_download / mdx_upload moves  [ref: md.atoms[i].posit/.force,
 * src/mol_alignment.rs:349-352; src/properties/sol_shrinking_box.rs:777-786] */
#define MDX_POS   0
#define MDX_VEL   1
#define MDX_FORCE 2

/* Flat SoA system description: what `MdState::new(dev,&cfg,&mols,param_set)` receives after the
 * host has parameterised its molecules  [ref: src/md/mod.rs:689, 1110-1151].  All pointers are
 * host memory; the handle copies everything, the caller may free immediately. */
typedef struct mdx_system {
    uint32_t n_atoms;
    const float*    pos;       /* [3N] Å, xyz interleaved                                         */
    const float*    vel;       /* [3N] Å/ps or NULL (zero)   [ref: atom_init_velocities, :1146]   */
    const float*    mass;      /* [N]  Da                                                         */
    const float*    charge;    /* [N]  e                                                          */
    const uint32_t* lj_type;   /* [N]  index into lj_sigma/lj_eps                                 */
    uint32_t        n_lj_types;
    const float*    lj_sigma;  /* [T]  Å          [ref: ff_params.rs:564-576 lennard_jones{sigma,eps}] */
    const float*    lj_eps;    /* [T]  kcal/mol                                                   */
    const uint8_t*  flags;     /* [N]  MDX_ATOM_* or NULL                                         */

    uint32_t        n_bonds;       /* E = k (r - r0)^2          [ref: ff_params.rs:352-372]       */
    const uint32_t* bond_idx;      /* [2*n_bonds]                                                 */
    const float*    bond_k;
    const float*    bond_r0;

    uint32_t        n_angles;      /* E = k (theta - theta0)^2  [ref: ff_params.rs:401-421]       */
    const uint32_t* angle_idx;     /* [3*n_angles]  i-j-k, j is the apex                          */
    const float*    angle_k;
    const float*    angle_theta0;  /* rad                                                         */

    uint32_t        n_dihedrals;   /* E = v [1 + cos(n phi - phase)], v = barrier_height/divider  */
    const uint32_t* dihedral_idx;  /* [4*n_dihedrals]  proper i-j-k-l; improper as listed          */
    const float*    dihedral_v;    /*                           [ref: ff_params.rs:474-511]       */
    const float*    dihedral_phase;/* rad                                                         */
    const int32_t*  dihedral_n;    /* periodicity                                                 */

    const uint32_t* excl_offsets;  /* [N+1] CSR of fully excluded partners (1-2, 1-3), symmetric  */
    const uint32_t* excl_idx;
    uint32_t        n_pairs14;     /* 1-4 pairs: excluded from the pair loop, evaluated scaled    */
    const uint32_t* pairs14_idx;   /* [2*n_pairs14]                                               */

    uint32_t        n_mols;        /* [ref: md.mol_start_indices, src/md/mod.rs:809-947]          */
    const uint32_t* mol_start;     /* [n_mols] first atom of each molecule, or NULL               */

    int32_t periodic;              /* 1: orthorhombic PBC in box_lo..box_hi; 0: vacuum            */
    float   box_lo[3];             /* [ref: SimBox{bounds_low,bounds_high}, sol_shrinking_box.rs:600-603] */
    float   box_hi[3];

    /* SURVEY §8f rank 1 — the reference's default operating point is dt = 2 fs with constrained
     * hydrogens and 4-site OPC water  [ref: HydrogenConstraint::{Shake,Linear,Flexible},
     * src/ui/panels/md.rs:362-371; md.water[i].{o,h0,h1,m}, sol_shrinking_box.rs:605-613]. */
    uint32_t        n_constraints;  /* holonomic distance constraints: X-H bonds; rigid water = O-H, O-H, H-H.
                                       Connected clusters span at most 4 atoms / 6 constraints, or are a heavy atom with four
                                       hydrogens (5 atoms, 4 bonds from one centre: ammonium, methane).
                                       HydrogenConstraint::Shake{shake_tolerance} -> constraint_tol.
                                       HydrogenConstraint::Linear{order, iter} (LINCS, the reference's UI default,
                                       src/ui/panels/md.rs:363-366) maps onto the SAME solver: LINCS's `order` (terms of
                                       the matrix expansion) and `iter` (correction passes) only bound its truncation
                                       error; this engine iterates every cluster in registers until each constraint is
                                       within constraint_tol (default 1e-5 relative; LINCS order 4 / iter 1 leaves ~1e-4),
                                       rigid three-site waters in closed form (SETTLE).  A host passes neither field;
                                       tests/test_gpu_constraints.py::test_methyl_clusters_meet_the_tolerance_linear_maps_to. */
    const uint32_t* constraint_idx; /* [2n] */
    const float*    constraint_len; /* [n] Å */
    uint32_t        n_vsites;       /* massless 3-parent virtual sites (OPC / TIP4P "M"):            */
    const uint32_t* vsite_idx;      /* [4n] site, p0, p1, p2;  r = r0 + a (r1 - r0) + b (r2 - r0)     */
    const float*    vsite_w;        /* [2n] a, b                                                      */
} mdx_system;

/* The subset of MdConfig the force/integrate path reads  [ref: src/ui/panels/md.rs:252-261,
 * 291-305; src/md/mod.rs:671-686].  Fill with mdx_config_default() first. */
typedef struct mdx_config {
    float    lj_cutoff;        /* Å; <=0 or inf: no cutoff (vacuum only)       [ref: md.rs:252-261] */
    float    coulomb_cutoff;   /* Å                                                               */
    float    skin;             /* Å Verlet buffer; lists rebuilt when max displacement > skin/2.  0 (periodic, single device) = the
                                  library chooses: it starts at 2 Å and walks to the skin with the best MEASURED step rate (fewer
                                  rebuilds against more listed pairs; forces do not depend on it).  mdx_get_skin reads it.   */
    float    coulomb_k;        /* 332.0637 kcal·Å/(mol·e²)                                        */
    float    scale14_lj;       /* 0.5     (Amber 1/2.0)                                           */
    float    scale14_coulomb;  /* 0.8333… (Amber 1/1.2)                                           */
    int32_t  coulomb_mode;     /* MDX_COULOMB_*                                                   */
    float    ewald_alpha;      /* 1/Å, MDX_COULOMB_EWALD only                                     */
    int32_t  combining_rule;   /* MDX_COMBINE_*                                                   */
    uint32_t overrides;        /* MDX_OVR_* bit set                                               */
    float    softening_sq;     /* Å², added to r² in the Coulomb force (src/cuda/util.cu:9 uses 1e-6); default 0 */
    uint32_t chunk_steps;      /* steps enqueued between host checks of the rebuild flag (mdx_config_default: 16; 0 = the
                                * library chooses by system size: 48 below 131 k atoms, else 16) */
    uint32_t nb_variant;       /* pair kernel (A/B knob): 0 = library default (5); 1 = whole-tile, full list; 2 = cluster-masked, full
                                  list, bitwise reproducible; 3/4 = 2 with 1/4 waves per tile; 5 = cluster-masked, HALF list, the
                                  reaction force written back with f32 atomics (fastest; last bits vary run to run) */
    float    constraint_tol;   /* relative tolerance of SHAKE/RATTLE (HydrogenConstraint::Shake{shake_tolerance}); default 1e-5 */
    uint32_t constraint_max_iter; /* default 64 */
    uint32_t pme_grid[3];      /* SPME mesh (MDX_COULOMB_EWALD without MDX_OVR_LONG_RANGE_RECIP_DISABLED);
                                  0 = smallest 2^a 3^b 5^c size with spacing <= 1 Å */
    uint32_t pme_order;        /* B-spline order: 4 (default) */
    float    inner_skin;       /* Å; dual pair list: the pair kernel walks an INNER list = the cluster pairs of the Verlet list with an
                                  atom pair inside cutoff + inner_skin, re-pruned on the device (inside the pair kernel, no host
                                  round trip) whenever an atom's path length since the last pruning exceeds inner_skin/2.
                                  0 = library default (0.5); < 0 or >= skin: off.  Results are those of the plain list. */
} mdx_config;

/* Superset of SnapshotEnergyData  [ref: src/ui/panels/md_viewer.rs:195-257; src/md/mod.rs:1241-1245] */
typedef struct mdx_energies {
    double kinetic;             /* kcal/mol */
    double potential;           /* = potential_nonbonded + potential_bonded */
    double potential_nonbonded; /* lj + coulomb + lj14 + coulomb14 */
    double potential_bonded;    /* bond + angle + dihedral */
    double lj, coulomb, lj14, coulomb14;
    double bond, angle, dihedral;
    double temperature;         /* K, 2 KE / (dof kB), dof = 3 N_mobile - 3 (>=1) */
    double volume;              /* Å³ (0 in vacuum) */
    double density;             /* amu/Å³ (0 in vacuum) */
    double virial;              /* W = sum_i r_i·F_i (kcal/mol) of all internal forces: pair, 1-4, bonded, SPME reciprocal
                                   and - with constraints - the SHAKE forces of the last step */
    double max_force;           /* max |F| kcal/mol/Å — blow-up detector (sol_shrinking_box.rs:776-789) */
    double coulomb_recip;       /* SPME reciprocal sum + self + excluded-pair + background terms (in
                                   potential_nonbonded); `coulomb` is the real-space part */
    double dh_dlambda;          /* kcal/mol: dU/dlambda of the alchemical window (0 when none is configured)       */
    double coupled_interaction; /* kcal/mol: (1 - lambda) x the non-bonded energy between the coupled molecule and the
                                   rest (mean_coupled_interaction_kcal's sample, src/properties/water_sol.rs:442)       */
    double pressure;            /* bar, (2 KE + W) / (3 V) x 69476.95; 0 in vacuum  [ref: en.pressure,
                                   src/ui/panels/md_viewer.rs:246; src/properties/crystal.rs:526] */
} mdx_energies;

/* Counters and timers (md.computation_time() analogue, src/md/mod.rs:740-743). */
typedef struct mdx_stats {
    uint64_t step_count;        /* [ref: md.step_count, src/md/mod.rs:738] */
    uint64_t rebuild_count;
    uint32_t n_atoms, n_slots, n_tiles, n_clusters;
    uint64_t n_list_entries;    /* (tile, j-cluster) entries in the current pair list */
    uint64_t n_masked_entries;
    uint64_t n_cluster_pairs;   /* (i-cluster, j-cluster) pairs within the list radius; x64 = pair evaluations per force call */
    /* HIP-event timing of the kernels, filled only while mdx_profile(h,1) is on */
    double   nb_ms_sum;    uint64_t nb_launches;
    double   bonded_ms_sum; uint64_t bonded_launches;
    double   integ_ms_sum;  uint64_t integ_launches;
    double   rebuild_ms_sum;
    double   wall_ms_sum;       /* host wall time inside mdx_step */
    uint64_t n_inner_cluster_pairs; /* dual list: cluster pairs the last pruning pass kept (0 when off) */
    uint64_t prune_passes;      /* dual list: pruning passes executed so far */
    /* decomposed handles (mdx_comm_init): this rank's share and its repartition history */
    uint32_t n_owned, n_ghost;
    uint64_t repartitions;      /* times the ranks re-derived owners / ghosts / halo lists from the gathered state */
    uint64_t local_rebuilds;    /* stale lists rebuilt on the unchanged owned + ghost set */
    double   repartition_ms_sum;
    /* step loop of the large classes: bonded gather + full kick + drift as one pass (not counted under bonded / integ) */
    double   fused_ms_sum;  uint64_t fused_launches;
    /* energies: force calls of the energy flavour enqueued so far (one the device gated off behind a stale list counts), and mdx_energy / snapshot / barostat reads served by the evaluation
     * the step loop made at a cadence step (mdx_set_energy_cadence) instead of one of their own */
    uint64_t energy_evaluations, energies_from_step_loop;
    /* list rebuilds that left the fused chain for the unfused one (a tile's entries beyond the LDS buffer of the single-pass build, a
     * list that outgrew its arrays): correct, 2-4 x slower - a count that keeps growing says the buffers are too small for this system */
    uint64_t rebuild_fallbacks;
} mdx_stats;

typedef struct mdx_handle mdx_handle;

/* Number of usable gfx950 devices; 0 lets the host keep its CPU path
 * [ref: get_computation_device, src/util.rs:1072-1119]. */
int mdx_device_count(void);

/* Thread-local description of the last failure  [ref: ParamError.descrip]. */
const char* mdx_last_error(void);

void mdx_config_default(mdx_config* cfg);

/* MdState::new  [ref: src/md/mod.rs:689; src/docking/mod.rs:235; src/mol_editor/mod.rs:901]. */
int mdx_create(const mdx_system* sys, const mdx_config* cfg, int device, mdx_handle** out);
void mdx_destroy(mdx_handle* h);

/* MdState::step(&dev, dt, Option<Vec<Vec3F32>>) repeated n_steps times on the device
 * [ref: src/md/mod.rs:716,748 (10-step GUI burst :737); src/mol_alignment.rs:346 (ext forces)].
 * ext_forces: [3N] kcal/mol/Å in caller atom order, held constant over the burst, or NULL. */
int mdx_step(mdx_handle* h, float dt, const float* ext_forces, uint32_t n_steps);

/* Forces + per-term energies of the current state without stepping (the per-snapshot
 * energy_data, src/md/viewer.rs:378-394). */
int mdx_energy(mdx_handle* h, mdx_energies* out);

/* dynamics::compute_energy_snapshot  [ref: src/md/mod.rs:1036]: stateless single-point scorer.
 * forces_or_null: [3N]. */
int mdx_single_point(const mdx_system* sys, const mdx_config* cfg, int device,
                     mdx_energies* out, float* forces_or_null);
/* The scorer is called pose after pose on the same molecules [ref: src/docking/mod.rs:235]: the calling thread keeps the
 * device state of its last mdx_single_point call, and a call whose system has the same static content (everything but
 * pos / vel / box, compared by a fingerprint of the arrays) and the same config only uploads the new coordinates
 * that changed since the last pose (mdx_upload_range: the Verlet list survives small ligand moves).  Results are
 * those of a fresh build.  This frees the kept state; a thread's kept state is also freed when the thread exits. */
void mdx_single_point_release(void);

/* State read-back / host-side mutation  [ref: md.atoms[i].posit/.force public fields]. */
int mdx_download(mdx_handle* h, int which, float* dst /* [3N] */);
int mdx_upload(mdx_handle* h, int which, const float* src /* [3N] */);
/* The docking loop moves ~50 ligand atoms of a ~50 k-atom complex from pose to pose [ref: src/docking/mod.rs:81-154,
 * 235]: new values for atoms [first, first + count) only.  For MDX_POS the spatial caches are KEPT while every moved
 * atom stays within skin/2 of where it was when the Verlet list was built (the same rule the step loop uses); beyond
 * that the list is rebuilt on next use.  Results are those of a fresh build.  mdx_upload (whole array) always
 * invalidates the caches, as `md.rebuild_spatial_caches()` after a host-side edit does. */
int mdx_upload_range(mdx_handle* h, int which, uint32_t first, uint32_t count, const float* src /* [3 count] */);

/* md.cell = SimBox::new(lo,hi) followed by md.rebuild_spatial_caches()
 * [ref: src/properties/sol_shrinking_box.rs:600-603, 632]. */
int mdx_set_box(mdx_handle* h, const float lo[3], const float hi[3]);
int mdx_rebuild_spatial_caches(mdx_handle* h);

uint64_t mdx_step_count(const mdx_handle* h);

/* Verlet neighbour list of the current spatial caches, in caller atom order: all j != i with
 * canonical fp32 minimum-image distance r2 < (max cutoff + skin)^2 at the positions of the last
 * rebuild, each row sorted ascending.  Call with idx == NULL to get offsets[N+1] only
 * (offsets[N] = total), then again with idx sized offsets[N]. */
int mdx_neighbor_list(mdx_handle* h, uint32_t* offsets /* [N+1] */, uint32_t* idx /* or NULL */);

/* (enable 3, decomposed handles: every phase of the step in its production arrangement - see mdx_comm_diag) */
/* HydrogenConstraint::{Flexible, Shake{shake_tolerance}, Linear{order, iter}} as the reference's UI hands it over
 * [ref: src/ui/panels/md.rs:362-371; LINCS_ORDER_DEFAULT, LINCS_ITER_DEFAULT, SHAKE_TOL_DEFAULT of the dynamics crate].
 * The constraints themselves are part of mdx_system; this call tells the solver how hard to work:
 *   MDX_HC_SHAKE   constraint_tol = shake_tolerance (relative; <= 0 keeps the configured one)
 *   MDX_HC_LINEAR  LINCS's order (terms of its matrix expansion) and iter (correction passes) bound a truncation error; this
 *                  engine solves every cluster in registers to a tolerance instead, so the pair is MAPPED: the tolerance
 *                  becomes min(configured, 10^-(order/2 + iter + 1)) - order 4 / iter 1 -> 1e-4, never looser than LINCS
 *                  itself would leave - and the mapping is written to mdx_constraint_description (nothing is ignored
 *                  silently)
 *   MDX_HC_FLEXIBLE refused (MDX_EPARAM) on a system created with constraints: build it without them. */
enum { MDX_HC_FLEXIBLE = 0, MDX_HC_SHAKE = 1, MDX_HC_LINEAR = 2 };
int mdx_set_hydrogen_constraint(mdx_handle* h, int kind, uint32_t lincs_order, uint32_t lincs_iter, float shake_tolerance);
/* One line: solver per cluster kind, tolerance in force, and how a Linear{order, iter} request was mapped. */
const char* mdx_constraint_description(mdx_handle* h);

/* Kernel timing with HIP events on the library's stream; mdx_get_stats reads the sums.  enable: 0 off, 1 every
 * step kernel, 2 the pair kernel only (each event pair is two extra packets in the queue: level 2 disturbs the
 * step loop least). */
int mdx_profile(mdx_handle* h, int enable);
int mdx_get_stats(mdx_handle* h, mdx_stats* out);
/* Diagnostics: which instantiation of the pair kernel the handle launched last - out[0..7] by the step loop over the dual pair
 * list, out[8..15] by any other force call (mdx_energy, the minimiser, single points); all zero until such a launch happened.
 * Per block: {waves per tile (0: the whole-tile kernel), dual-list body (0 plain list, 1 / 2 inner-walk / pruning twins, 3 one merged
 * launch - the device picks the body, 4 merged + the bonded gather in extra workgroups, 5 merged + the previous step's kick and drift
 * and the bonded roles of the tile's atoms inside: one launch per step), half list, Coulomb flavour (0 shifted cutoff,
 * 1 reaction field, 2 Ewald closed form, 3 softened, 4 Ewald table), energy flavour, workgroups per tile, bonded workgroups behind
 * twin launches, tiles in the launch}.  out[16]: list rebuilds so far whose exact pruning pass also wrote the inner list (the force
 * call behind such a rebuild walks the inner list instead of being a pruning pass: one wave per tile, single device), out[17]: the
 * last rebuild was one of them, out[18]: steps so far that took the handle's rigid waters through the one-pass water_step_kernel, out[19]: ... of
 * which with other mobile atoms beside them (a solute in rigid water), out[20]: pair launches so far that were the whole step (body 5),
 * out[21]: steps taken back to a list rebuild because a kick exceeded what the gating words had granted it, out[22..23]: 0.  The parity tests use it to name the body they hold against the oracle; no call
 * of the reference corresponds to it. */
int mdx_pair_launch_info(const mdx_handle* h, uint32_t out[24]);
/* The Verlet skin in force, and whether the library is still tuning it (mdx_config.skin == 0). */
int mdx_get_skin(const mdx_handle* h, float* skin, int* tuning);

/* ---- callers either side of `step` -----------------------------------------------
 * The reference reaches these through the same MdState: `md.minimize_energy(dev, iters, ext)`
 * (src/ui/mol_editor.rs:375; src/mol_alignment.rs:356; sol_shrinking_box.rs:962),
 * `md.initialize_velocities(TEMP, zero_com_drift)` (sol_shrinking_box.rs:965),
 * `Integrator::VerletVelocity{thermostat: Some(tau)}` + `temp_target` (src/ui/panels/md.rs:296-305;
 * CSVR per README.md:237-238), `zero_com_drift` (water_sol.rs:144), `snapshot_handlers.memory:
 * Some(every_n)` (water_sol.rs:185-189), `md.flush_snapshot_queues()` + `md.snapshots`
 * (src/md/mod.rs:118-122).  Their algorithms live in the absent crate; the ones implemented here
 * are stated in DESIGN.md §8 and restated by the oracle. */
#define MDX_THERMOSTAT_NONE      0
#define MDX_THERMOSTAT_BERENDSEN 1 /* lambda^2 = 1 + (Dt/tau)(T0/T - 1)                              */
#define MDX_THERMOSTAT_CSVR      2 /* Bussi-Donadio-Parrinello stochastic velocity rescaling         */

/* `Integrator::{VerletVelocity{thermostat}, Leapfrog{thermostat}, LangevinMiddle{gamma}}`
 * (src/ui/panels/md.rs:296-305, README.md:237-238).  Velocity Verlet is the default.  Leapfrog keeps
 * half-step velocities: v(t+dt/2) = v(t-dt/2) + dt a(t), x += dt v.  Langevin "middle" (Zhang et al. 2019):
 * v += dt a;  x += dt/2 v;  v = a1 v + sqrt(kT (1 - a1^2)/m) xi, a1 = exp(-gamma dt);  x += dt/2 v, with
 * xi three normals per atom and step from a counter-based stream (seed, step, atom id), so a trajectory does
 * not depend on how steps are batched or atoms are ordered on the device.  For both, velocities read back
 * (and the kinetic energy / temperature reported) are the half-step ones. */
#define MDX_INTEGRATOR_VERLET_VELOCITY 0
#define MDX_INTEGRATOR_LEAPFROG        1
#define MDX_INTEGRATOR_LANGEVIN_MIDDLE 2
int mdx_set_integrator(mdx_handle* h, int kind, float gamma_per_ps, float temperature, uint64_t seed);

/* `md.configure_alchemical_window(dev, mol_index, lambda)` (src/properties/water_sol.rs:556): thermodynamic
 * integration windows, lambda from 0 (full interaction between the molecule and its environment) to 1 (none); the
 * reference's grid ends ... 0.90, 0.95, 1.0 (water_sol.rs:52-56) and runs under its default SPME Coulomb (README.md:240).
 * The coupling form lives in the absent crate; built here: the non-bonded interactions between the atoms of molecule
 * `mol_index` and every other atom are scaled by (1 - lambda) and evaluated at the SOFT-CORE distance
 *     r_sc = (alpha sigma_ij^6 lambda + r^6)^(1/6)        (Beutler et al. 1994; alpha 0.5; sigma_min 3 A for pairs without LJ)
 * - LJ and real-space Coulomb alike - so dU/dlambda stays finite at lambda = 1 whatever overlaps the decoupled molecule;
 * U(lambda) = U_rest + (1 - lambda) U_cross(r_sc).  Intramolecular terms (bonded, 1-4, intra non-bonded) are not scaled.
 * With the SPME reciprocal sum the mesh part of the cross interaction is scaled exactly: environment and molecule are
 * spread and transformed separately (the mesh potential is linear in the charges), E_rec(lambda) = E_env,env + E_mol,mol
 * + (1 - lambda) 2 E_env,mol, and each atom's force is interpolated from its own group's potential plus (1 - lambda)
 * times the other's; the uniform background of a charged molecule is left unscaled.
 * Every energy evaluation (hence every snapshot) reports dh_dlambda = dU/dlambda and coupled_interaction =
 * (1 - lambda) U_cross.  Needs mol_start in the system description.  Works on a decomposed handle too (cutoff and SPME; the
 * reciprocal mesh is then replicated and all-reduced instead of slab-decomposed: mdx_pme_info).
 * lambda < 0 switches the window off.  mdx_set_alchemical_softcore: alpha = 0 selects plain linear coupling
 * (singular in dU/dlambda at lambda -> 1 for overlapping sites). */
int mdx_set_alchemical_softcore(mdx_handle* h, float alpha, float sigma_min);
int mdx_configure_alchemical_window(mdx_handle* h, uint32_t mol_index, double lambda);

#define MDX_BAROSTAT_NONE      0
#define MDX_BAROSTAT_BERENDSEN 1 /* mu^3 = 1 - compressibility (Dt/tau) (P0 - P); box and coordinates scaled by mu */
#define MDX_BAR_PER_KCAL_MOL_A3 69476.95

/* Steepest descent with an adaptive maximum displacement (start 0.01 Å; x += h F/|F|max; accepted
 * when the potential energy drops: h *= 1.2, else the move is undone and h *= 0.5); stops after
 * max_iters force evaluations or when max |F| < f_tol; h never exceeds 0.2 Å.  Velocities are left untouched.
 * With external forces (the alignment pull, src/mol_alignment.rs:356) the accepted quantity is U - sum F_ext . x
 * (the internal potential minus the work of the external forces along the move), so the molecule follows the pull. */
int mdx_minimize_energy(mdx_handle* h, uint32_t max_iters, const float* ext_forces_or_null, float f_tol,
                        mdx_energies* final_or_null, uint32_t* iters_done_or_null);
/* Maxwell-Boltzmann velocities from a counter-based generator (splitmix64 + Box-Muller, three
 * normals per atom in caller order) so that a given seed means the same velocities everywhere. */
int mdx_initialize_velocities(mdx_handle* h, float temperature, int zero_com_drift, uint64_t seed);
/* Velocity rescaling applied every `every_n_steps` steps (coupling interval Dt = every_n_steps*dt). */
/* `BarostatCfg{tau, pressure_target}` (src/ui/panels/md.rs:517-557, src/properties/crystal.rs:312-315):
 * weak-coupling (Berendsen-style) isotropic pressure control applied every `every_n_steps` steps: the
 * pressure of the current state is evaluated (one energy-flavoured force pass), box edges and atom
 * coordinates are scaled about box_lo by mu = cbrt(1 - compressibility (Dt/tau) (P0 - P)) (|mu - 1| capped
 * at 1 %), the pair list is rebuilt and constrained clusters are re-projected.  Periodic systems on one
 * device only; compressibility <= 0 selects water's 4.5e-5 / bar. */
/* `md.shrink_cell_towards(dev, target_cell, ShrinkingBoxCfg{box_shrink_per_step, ..}) -> bool`
 * [ref: src/properties/sol_shrinking_box.rs:990, called once per MD step of the packing run].  Every edge of the
 * cell shrinks by shrink_per_step Å but not below the target's edge, the cell keeps its centre (the rule the
 * reference states at sol_shrinking_box.rs:765-774); coordinates follow affinely about the centre, velocities are
 * untouched, the spatial caches are rebuilt by the next step and constrained clusters re-projected.  *shrank_out
 * (may be NULL) = 1 when any edge changed.  MDX_EPARAM (state untouched) when an edge would drop below
 * 2 (cutoff + skin).  Fully periodic, single-device handles. */
int mdx_shrink_cell_towards(mdx_handle* h, const float target_lo[3], const float target_hi[3], float shrink_per_step,
                            int* shrank_out);
/* md.cell read-back: the current box (changes under the barostat). */
int mdx_get_box(const mdx_handle* h, float lo[3], float hi[3]);
int mdx_set_barostat(mdx_handle* h, int kind, float pressure_target_bar, float tau_ps, float compressibility_per_bar,
                     uint32_t every_n_steps);
int mdx_set_thermostat(mdx_handle* h, int kind, float temp_target, float tau_ps, uint32_t every_n_steps,
                       uint64_t seed);
int mdx_set_zero_com_drift(mdx_handle* h, int enable);   /* removed at the thermostat cadence (or every 100 steps) */
/* Energies at a cadence.  The reference reads them at the ratio of its snapshot handlers (`SnapshotHandlers`, /root/reference
 * src/md/mod.rs:121-122; src/properties/water_sol.rs:185-189), i.e. the caller knows in advance at which steps it will ask.  Told
 * that cadence, mdx_step evaluates the energies as part of the force call that ends every step whose count is a multiple of
 * `every_n_steps` (the energy flavour of the same kernels, on the same positions), and mdx_energy called at such a step
 * returns them without a second evaluation; at any other step, or after anything has changed the state, mdx_energy evaluates
 * afresh as before.  The snapshot cadence and the barostat's coupling interval get the same treatment by themselves.
 * 0 = off (default). */
int mdx_set_energy_cadence(mdx_handle* h, uint32_t every_n_steps);
/* In-memory snapshots every `every_n` steps: time, step, energies, positions (and velocities). */
int      mdx_set_snapshot_cadence(mdx_handle* h, uint32_t every_n, int with_velocities);
uint32_t mdx_snapshot_count(const mdx_handle* h);
int      mdx_snapshot_read(mdx_handle* h, uint32_t k, double* time_ps, uint64_t* step, mdx_energies* e,
                           float* pos /* [3N] */, float* vel_or_null /* [3N] */);
int      mdx_flush_snapshot_queues(mdx_handle* h);       /* drop the stored snapshots (after the host cloned them) */
/* `MdConfig.snapshot_handlers { memory: Option<every_n>, dcd: Option<every_n>, gromacs: OutputControl { nstxout, nstvout, nstfout,
 * nstenergy, nstcalcenergy, nstxout_compressed } }`  [ref: src/properties/water_sol.rs:185-189; src/properties/crystal.rs:335-342;
 * src/ui/panels/md.rs:775-899]: every handler with its own cadence, below the ABI.  A snapshot is stored after every step that is a
 * multiple of ANY handler's cadence (0 = handler off); it always carries time, step, energies and positions, velocities when the
 * nstvout handler (or mdx_set_snapshot_cadence's with_velocities) is due, forces when nstfout is due; mdx_snapshot_handler_mask says
 * which handlers wanted it (bit MDX_SNAP_*; bit 31: the plain cadence of mdx_set_snapshot_cadence), so the host feeds each of its
 * writers - the in-memory list, the DCD / TRR / XTC / EDR files, which stay on the Rust side - exactly the frames it asked for.
 * Energies of the steps nstenergy / nstcalcenergy name are evaluated inside the step loop (as with mdx_set_energy_cadence). */
#define MDX_SNAP_HANDLERS 8
enum { MDX_SNAP_MEMORY = 0, MDX_SNAP_DCD = 1, MDX_SNAP_NSTXOUT = 2, MDX_SNAP_NSTVOUT = 3, MDX_SNAP_NSTFOUT = 4, MDX_SNAP_NSTENERGY = 5,
       MDX_SNAP_NSTCALCENERGY = 6, MDX_SNAP_NSTXOUT_COMPRESSED = 7 };
int      mdx_set_snapshot_handlers(mdx_handle* h, const uint32_t every_n[MDX_SNAP_HANDLERS] /* NULL: all off */);
uint32_t mdx_snapshot_handler_mask(const mdx_handle* h, uint32_t k);
int      mdx_snapshot_read_forces(mdx_handle* h, uint32_t k, float* frc /* [3N] */);

/* ---- md.water and the water / hydrogen-bond part of a Snapshot -------------------------------------------------------
 * The reference keeps solvent water apart from `md.atoms`: `md.water[i].{o,h0,h1,m}.{posit,force}`
 * [ref: src/properties/sol_shrinking_box.rs:605-613, 780-786], and a Snapshot carries `atom_posits` (the non-water
 * atoms), `water_o_posits / water_h0_posits / water_h1_posits` and, in its energy data, `hydrogen_bonds`
 * [ref: src/md/viewer.rs:374-394, 917-960; src/properties/water_sol.rs:268-300].  Here the system is one flat atom
 * array; the host says where its waters are - n_waters contiguous records of `sites_per_water` atoms in the order
 * O, H0, H1 (, M) starting at first_atom - and the library hands out the reference's views:
 *   mdx_water_download      md.water[i].{o,h0,h1,m}.posit / .force of the current state, [3 n_waters] floats each; positions
 *                           come out with every molecule WHOLE (H0, H1, M in the periodic image next to their O: the flat
 *                           array of mdx_download keeps each atom wrapped into the cell on its own); the same holds for
 *                           mdx_snapshot_read_water
 *   mdx_snapshot_read_water water_o/h0/h1_posits of stored snapshot k (atom_posits = rows [0, first_atom) of its pos)
 * Hydrogen bonds: the detection rule lives in the absent crates (bio_files::bond_inference::h_bond_geometry_strength);
 * built here, switchable per handle: a hydrogen bound (bond or constraint) to a heavy atom flagged in
 * `is_hbond_heavy` (N, O, S, F: the reference's candidate elements, water_sol.rs:198-203) donates to another flagged
 * atom when H...A <= max_h_acc_dist (default 2.5 A) and the angle D-H...A >= min_angle_deg (default 120), minimum
 * image; strength = (1 - (d - 1.5)/(max - 1.5)) x (angle - min)/(180 - min), clamped to [0, 1].  Detected at every
 * snapshot (host side, linear-time cell search) once switched on.  Atom references follow the reference's
 * `(HBondAtomType, index)`: MDX_HB_STANDARD indexes the non-water atoms, the water types index the water. */
#define MDX_HB_STANDARD 0
#define MDX_HB_WATER_O  1
#define MDX_HB_WATER_H0 2
#define MDX_HB_WATER_H1 3
typedef struct mdx_hbond {
    uint32_t donor, acceptor, hydrogen;                 /* indices (see *_type) */
    uint8_t  donor_type, acceptor_type, hydrogen_type;  /* MDX_HB_* */
    uint8_t  pad;
    float    strength;
} mdx_hbond;
int      mdx_set_water_layout(mdx_handle* h, uint32_t first_atom, uint32_t n_waters, uint32_t sites_per_water);
int      mdx_water_download(mdx_handle* h, int which /* MDX_POS | MDX_FORCE */, float* o, float* h0, float* h1,
                            float* m_or_null /* 4-site water only */);
int      mdx_set_hbond_detection(mdx_handle* h, const uint8_t* is_hbond_heavy /* [N], NULL = off */, float max_h_acc_dist,
                                 float min_angle_deg);
int      mdx_snapshot_read_water(mdx_handle* h, uint32_t k, float* o, float* h0, float* h1 /* [3 n_waters] each */);
uint32_t mdx_snapshot_hbond_count(const mdx_handle* h, uint32_t k);
int      mdx_snapshot_read_hbonds(mdx_handle* h, uint32_t k, mdx_hbond* out, uint32_t capacity);
double   mdx_time_ps(const mdx_handle* h);

/* ---- SnapshotEnergyData.energy_potential_between_mols --------------------------------------------------------------------
 * [ref: src/properties/crystal.rs:347-370 `cohesive_energy_from_matrix(&e.energy_potential_between_mols, n_mol)`: flat row-major
 * n_mol x n_mol, called on every snapshot at :533; src/ui/panels/md_viewer.rs:234-237.]  The non-bonded energy between molecules -
 * or between caller-chosen groups of atoms: "receptor / ligand / solvent" makes it the docking scorer's receptor-ligand
 * interaction energy (BASELINE config 3, src/docking/mod.rs:81-154).  Off until groups are set:
 *   mdx_set_energy_groups(h, NULL, 0)        one group per molecule of mdx_system.mol_start (at most 255 molecules)
 *   mdx_set_energy_groups(h, group_of_atom, n) group_of_atom[i] < n <= 255 for every atom
 *   mdx_set_energy_groups(h, NULL, MDX_GROUPS_OFF)  off again, on any system (snapshots stop paying the extra pair-list pass);
 *                                            (h, NULL, 0) on a system without mol_start also means off
 * The matrix is symmetric, kcal/mol:
 *   M[a][b] (a != b)  sum over atom pairs (i in a, j in b) of the pair loop's Lennard-Jones + Coulomb energy - the configured
 *                     real-space treatment (shifted cutoff / reaction field / erfc(beta r)/r), same cutoffs, exclusions and periodic
 *                     images as the forces - plus the scaled 1-4 energy of 1-4 pairs between the two groups
 *   M[a][a]           the same over the pairs inside group a, every pair once
 * so that  sum_{a <= b} M[a][b] = lj + coulomb + lj14 + coulomb14  of mdx_energies (= potential_nonbonded - coulomb_recip: the SPME
 * mesh term belongs to the whole charge density and is not split).  With an alchemical window the coupled pairs enter scaled,
 * as in mdx_energies.  mdx_energy_between_mols evaluates the current state (one extra pass over the pair list; collective on a
 * decomposed handle); while groups are set every stored snapshot carries its matrix (mdx_snapshot_read_between_mols). */
#define MDX_GROUPS_OFF 0xFFFFFFFFu
int      mdx_set_energy_groups(mdx_handle* h, const uint8_t* group_of_atom /* [N] or NULL */, uint32_t n_groups);
uint32_t mdx_energy_group_count(const mdx_handle* h);
int      mdx_energy_between_mols(mdx_handle* h, float* out /* [n * n] row-major */, uint32_t n);
int      mdx_snapshot_read_between_mols(mdx_handle* h, uint32_t k, float* out /* [n * n] */, uint32_t n);
/* compute_energy_snapshot with the matrix: mdx_single_point (same pose cache, same results) followed by the matrix of the same
 * pose for the given groups (group_of_atom NULL: by molecule, and n_groups must then equal mdx_system.n_mols - matrix_out is
 * n_groups x n_groups either way; a mismatch is MDX_EPARAM, nothing is written).  The ligand row of a receptor / ligand / solvent map is what a
 * docking pose is ranked by. */
int      mdx_single_point_between_mols(const mdx_system* sys, const mdx_config* cfg, int device, const uint8_t* group_of_atom,
                                       uint32_t n_groups, mdx_energies* out, float* forces_or_null, float* matrix_out /* [n * n] */);

/* ---- multi-GPU: one periodic box spatially decomposed over the GPUs of a node (the reference is
 * single-device, src/util.rs:1086 `CudaContext::new(0)`, so this is new capability, not parity) -------------------
 * One rank (process or thread) per GPU.  Every rank creates a handle from the SAME global system (static per-atom data
 * and topology are replicated), then joins the communicator; from then on
 *     mdx_step      runs the decomposed step loop: drift -> pack -> ncclGroupStart / ncclSend + ncclRecv per peer /
 *                   ncclGroupEnd on the handle's communication stream -> unpack -> forces, the "list went stale" word
 *                   riding on the halo message; stale lists are rebuilt locally or the ranks repartition (decided alike
 *                   everywhere), all below this ABI
 *     mdx_energy    returns the totals of the whole box on every rank
 *     mdx_download  gathers the global array on every rank
 * and these three are COLLECTIVE: every rank must make the same calls in the same order.  Constraints, virtual sites,
 * thermostats, every integrator, external forces, snapshots, the SPME reciprocal sum (the mesh cut into x-slabs, one per rank: see mdx_pme_info), the
 * barostat (the all-reduced pressure gives every rank the same scale factor; the gathered state and the box are scaled alike,
 * then the ranks repartition), mdx_minimize_energy (collective: all-reduced energies and largest force drive ONE step-length
 * control; the accepted state is the gathered global positions) and alchemical windows (cutoff and SPME) work on a decomposed
 * handle (a constraint cluster / virtual-site family is owned as a whole by one rank).  The halo is a HALF shell when the
 * half-list pair kernel runs (the default): a pair of atoms owned by two ranks is evaluated on one of them and the force on
 * the ghost travels back in a second send/recv group per step.  Host mutation of a joined handle - mdx_upload, mdx_upload_range,
 * mdx_set_box, mdx_shrink_cell_towards, mdx_initialize_velocities - is COLLECTIVE as well: every rank passes the same data; the
 * distributed state is gathered, the named rows / the box are overwritten on every rank alike and the ranks repartition
 * (about 1 ms at 1 M atoms; a pose loop that wants the single-GPU latency keeps its handle undecomposed).
 *
 * mdx_comm_unique_id: rank 0 draws the id (ncclGetUniqueId; librccl is dlopen'd on first use) and hands the 128 bytes
 * to the other ranks by whatever means the host has.  mdx_comm_init is ncclCommInitRank + the first partition. */
#define MDX_COMM_ID_BYTES 128
int mdx_comm_unique_id(uint8_t id[MDX_COMM_ID_BYTES]);
int mdx_comm_init(mdx_handle* h, const uint8_t id[MDX_COMM_ID_BYTES], int rank, int world);
/* The same decomposition inside ONE process: `world` handles (one thread each; on one device or several) meet through
 * an in-process fabric instead of RCCL - plain device-to-device copies between their halo buffers.  What single-GPU
 * test boxes use (RCCL refuses two ranks on one device) and what a single-process host may use. */
typedef struct mdx_fabric mdx_fabric;
mdx_fabric* mdx_fabric_create(int world);
void mdx_fabric_destroy(mdx_fabric* f);
void mdx_fabric_abort(mdx_fabric* f);          /* a rank failed: release everybody who waits for it */
int mdx_comm_init_fabric(mdx_handle* h, mdx_fabric* f, int rank);
/* The same decomposition between PROCESSES of one host without RCCL: rows are staged through a POSIX shared-memory segment
 * `/mdx_<name>` (every rank passes the same name; rank 0 creates it; slots of MDX_SHM_SLOT_MB = 64 MiB per rank).  Not a
 * performance path - it runs the process-per-rank flow on a box whose single GPU RCCL will not share between two ranks. */
int mdx_comm_init_shm(mdx_handle* h, const char* name, int rank, int world);
/* Rank `rank` of `world` with a transport that delivers nothing: what ONE rank of a decomposition costs per step,
 * measured alone on a single GPU (tools/one_rank_profile.py). */
int mdx_comm_init_null(mdx_handle* h, int rank, int world);
/* Diagnostics: every transport entry point of a joined handle on its real wire (send/recv group to every rank incl. itself,
 * the small and the large all-reduce, the word all-gather), results checked.  Collective. */
int mdx_comm_selftest(mdx_handle* h);
/* Diagnostics of the error path: issues a send to a rank that does not exist INSIDE a send/recv group.  Returns MDX_OK
 * when the transport behaved - it reported the failure, closed its group (no later call hangs in an open group) and
 * refuses further traffic - and MDX_EDEVICE when it did not.  The handle's communicator is unusable afterwards:
 * destroy the handle.  Not collective (nothing reaches the wire). */
int mdx_comm_selftest_fault(mdx_handle* h);
int mdx_comm_info(const mdx_handle* h, int* rank, int* world, int grid[3], uint32_t* n_owned, uint32_t* n_ghost, float* halo);

/* Where a decomposed step spends its time and what it moves: what a reader of one multi-GPU bench line needs to tell the
 * wire from the kernels.  Phase times are GPU time between HIP events on the stream each phase runs on, summed while
 * mdx_profile(h, 3) is on - level 3 keeps the production arrangement (interior tiles beside the message, fused passes) and
 * brackets every phase; phase_n counts the brackets.  The halo "wire" phase is the ncclSend/ncclRecv group (or the other
 * transports' copies): it includes waiting for the peers to arrive.  Not collective. */
#define MDX_DIAG_PHASES 10
enum { MDX_PHASE_HALO_PACK = 0, MDX_PHASE_HALO_WIRE = 1, MDX_PHASE_HALO_UNPACK = 2, MDX_PHASE_FORCE_PACK = 3,
       MDX_PHASE_FORCE_WIRE = 4, MDX_PHASE_FORCE_ADD = 5, MDX_PHASE_PAIR = 6 /* whole launch, or the interior half */,
       MDX_PHASE_PAIR_BOUNDARY = 7, MDX_PHASE_BONDED = 8 /* + the fused bonded + kick + drift pass */, MDX_PHASE_INTEGRATE = 9 };
typedef struct mdx_comm_diag {
    char     transport[64];          /* "rccl", "in-process fabric", "shared memory (host-staged)", "null (delivers nothing)" */
    int32_t  rank, world, grid[3];
    int32_t  rccl_version;           /* ncclGetVersion (e.g. 22606), 0 on the other transports */
    int32_t  rccl_comm_count;        /* ncclCommCount of this handle's communicator: must equal world */
    int32_t  half_shell;             /* 1: every cross-rank pair on one rank, ghost forces travel back */
    int32_t  overlap_split;          /* interior / boundary split of the pair kernel: 1 kept, 0 dropped, -1 still measuring */
    int32_t  comm_stream_separate;   /* 1: RCCL calls on their own stream (MDX_COMM_STREAM=1) */
    int32_t  wire_ns_measured;       /* one send/recv group of a typical halo message, measured when the handle joined (largest over the
                                      * ranks), in ns; -1: not measured (MDX_HALF_SHELL pinned, or a transport without a wire).  It decides
                                      * half_shell: two messages per step pay below ~8 us per message only */
    uint32_t n_owned, n_ghost, n_tiles, n_interior_tiles;
    uint32_t halo_rows_out, halo_rows_in;     /* float4 rows per position message, flag rows included */
    uint64_t halo_bytes_per_step;    /* positions out + in, plus the force rows back and forth with the half shell */
    uint64_t repartitions, local_rebuilds;
    double   repartition_ms_sum;
    double   phase_ms[MDX_DIAG_PHASES]; uint64_t phase_n[MDX_DIAG_PHASES];
} mdx_comm_diag;
int mdx_comm_diag_read(mdx_handle* h, mdx_comm_diag* out);
/* Diagnostics: what the partition kernels derived at the last (re)partition, per GLOBAL atom - class here (0 absent, 1 owned,
 * 2 ghost, 3 ghost kept only as a bonded partner), owning rank, image code ((kx+1) | (ky+1) << 2 | (kz+1) << 4) and, for owned
 * atoms, the bit set of ranks that keep a copy - and the two halo lists (global atom ids in message order, 0xFFFFFFFF = a
 * peer segment's flag row).  Any pointer may be NULL.  tests/test_gpu_partition_spec.py holds them against the executable
 * specification tests/decomp_spec.py. */
/* Diagnostics of the reciprocal-space mesh of a decomposed handle: slab_on = 1 when the mesh is cut into x-slabs (one per rank:
 * block -> slab redistribution, 2-D FFT + transpose + 1-D FFT; mdx_pme.hip), 0 when it is replicated and all-reduced (fallback: mesh
 * edges not divisible by the rank count, alchemical window, MDX_PME_SLAB=0); bytes this rank SENDS per force call for the
 * real-space mesh (charges to the slab owners + potential back to the block owners) and for the two FFT transposes. */
int mdx_pme_info(const mdx_handle* h, int* slab_on, uint64_t* mesh_bytes_sent, uint64_t* transpose_bytes_sent, uint64_t* replicated_mesh_bytes);
/* Diagnostic of the charge spread of a single-GPU handle (mesh cut into bricks, one bucket of atoms per brick; mdx_pme.hip): the
 * number of atoms, summed over all force calls so far, that did not fit their brick's bucket and went through the overflow list
 * (correct, slower).  0 on a handle without SPME or with another spread.  Waits for the device. */
int mdx_pme_brick_overflows(mdx_handle* h, uint64_t* n);
int mdx_comm_debug_partition(mdx_handle* h, uint8_t* cls, uint8_t* owner, uint8_t* image_code, uint32_t* send_mask /* [N] each */,
                             uint32_t* n_send, uint32_t* n_recv, uint32_t* send_ids, uint32_t* recv_ids, uint32_t capacity);

/* ---- the building blocks underneath (kept for hosts that drive the decomposition themselves, and for the tests) ----
 * One handle per GPU/rank, created from the GLOBAL system (static per-atom data and topology are
 * replicated: 288 GB of HBM per GPU make that free), then told which atoms it simulates:
 * its owned atoms plus ghost copies of every atom within cutoff+skin of its brick.  In a
 * decomposed dimension the local region is not periodic: ghosts arrive already shifted into the
 * rank's frame.  All d_* pointers are DEVICE memory; everything runs on mdx_stream(h). */
#define MDX_PERIODIC_NONE 0
#define MDX_PERIODIC_XYZ  1
#define MDX_PERIODIC_DIMS(x, y, z) (0x10 | ((x) ? 1 : 0) | ((y) ? 2 : 0) | ((z) ? 4 : 0))

int mdx_set_local_atoms(mdx_handle* h, uint32_t n_local, const uint32_t* d_gid /* [n] global ids */,
                        const uint8_t* d_ghost /* [n] 1 = halo copy */, const float* d_pos4 /* [4n] */,
                        const float* d_vel4 /* [4n] */, const float lo[3], const float hi[3] /* local extent */,
                        int32_t periodic /* MDX_PERIODIC_* of the LOCAL region */);
int mdx_local_state(mdx_handle* h, float* d_pos4, float* d_vel4);   /* local order, 4 floats per atom */

/* A chunk of steps without host synchronisation.  flag word s+1 is raised by the drift of step s
 * when an owned atom moved more than skin/2 since the last rebuild; the host all-reduces (max) that
 * word across ranks on the same stream, and every later kernel of the chunk gates on it. */
int      mdx_chunk_begin(mdx_handle* h);                                  /* zero the flag words   */
int      mdx_chunk_integrate(mdx_handle* h, int mode, float dt, uint32_t s); /* mode 0: half kick+drift,
                                                 1: full kick+drift, 2: closing half kick; gated on flag[s] */
int      mdx_chunk_forces(mdx_handle* h, int32_t s);                      /* gated on flag[s+1]; s<0: ungated */
int      mdx_chunk_end(mdx_handle* h, uint32_t n_words, uint32_t* flags_out); /* sync, copy flag words  */
void*    mdx_flag_words(mdx_handle* h);          /* device uint32[66]: the words the host all-reduces   */
uint32_t mdx_stale_threshold(const mdx_handle* h); /* a word above this (as uint) means "list is stale"  */
int      mdx_add_steps(mdx_handle* h, uint32_t n);

/* Halo traffic: gather owned positions by global id into a send buffer / scatter received ghost
 * positions (plus a per-ghost image shift) back.  4 floats per row.  A row whose id is 0xFFFFFFFF
 * is a FLAG row: pack writes the bit pattern of flag word `flag_word` into it, unpack merges (max)
 * a received one into the local word — when every rank is every other rank's peer (2, 4, 8 GPUs)
 * the stale-list decision thus rides on the halo message and needs no separate all-reduce.
 * flag_word < 0 disables that. */
int   mdx_pack_positions(mdx_handle* h, const uint32_t* d_gid, uint32_t n, float* d_out4, int32_t flag_word);
int   mdx_unpack_positions(mdx_handle* h, const uint32_t* d_gid, uint32_t n, const float* d_in4,
                           const float* d_shift4_or_null, int32_t flag_word);
void* mdx_stream(mdx_handle* h);                           /* hipStream_t                        */

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
