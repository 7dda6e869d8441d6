// mdx.hpp — C++17 host-side mirror of the `dynamics` surface Molchanica calls, over the C ABI of mdx.h.
//
// The reference's host is Rust and calls the engine through values and methods of `MdState`
// (/root/reference src/md/mod.rs:689-750, 1036; src/mol_editor/mod.rs:388,901; src/mol_alignment.rs:346-356;
// src/properties/sol_shrinking_box.rs:600-632, 962-995).  There is no Rust toolchain in the build image, so the
// compiled-language host above the ABI is this header: same names, argument meaning and error behaviour, RAII
// instead of Rust ownership (`MdState` is move-only and frees the device state when dropped, like the
// `Option<MdState>` the application stores, src/md/mod.rs:52,960).  INTEGRATION.md has the Rust binding.
//
//   auto md = mdx::MdState::create(system, cfg);         // MdState::new(dev, &cfg, &mols, param_set)
//   md.step(0.002f, nullptr, 10);                        // md.step(&dev, dt, None) x 10   (the GUI burst)
//   auto e = md.energy();                                // per-snapshot energy_data
//   mdx::compute_energy_snapshot(system, cfg);           // dynamics::compute_energy_snapshot
//
// Errors: `ParamError{descrip}` (src/md/mod.rs:967) for bad input, `DeviceError` for HIP failures and missing
// GPUs (the reference degrades to its CPU path, src/util.rs:1072-1119; this library has none and says so),
// `BlowUpError` for non-finite state (sol_shrinking_box.rs:776-789).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

extern "C" {
#include "mdx.h"
}

namespace mdx {

struct ParamError : std::invalid_argument {
    explicit ParamError(const std::string& descrip) : std::invalid_argument(descrip) {}
};
struct DeviceError : std::runtime_error {
    explicit DeviceError(const std::string& what) : std::runtime_error(what) {}
};
struct BlowUpError : std::runtime_error {
    explicit BlowUpError(const std::string& what) : std::runtime_error(what) {}
};

inline void check(int rc) {
    if (rc == MDX_OK) return;
    const char* m = mdx_last_error();
    const std::string msg = m ? m : "";
    if (rc == MDX_EPARAM) throw ParamError(msg);
    if (rc == MDX_ENAN) throw BlowUpError(msg);
    throw DeviceError(msg);   // MDX_EDEVICE, MDX_EOOM
}

/// `get_computation_device` (src/util.rs:1072-1119): number of usable gfx950 devices; 0 = keep the CPU path.
inline int device_count() { return mdx_device_count(); }

inline mdx_config default_config() {
    mdx_config c;
    mdx_config_default(&c);
    return c;
}

/// `Snapshot` (src/md/viewer.rs:378-394): time, positions, optional velocities, energy_data.
struct Snapshot {
    double time_ps = 0.0;
    uint64_t step = 0;
    mdx_energies energy_data{};
    std::vector<float> atom_posits;       // [3N]
    std::vector<float> atom_velocities;   // [3N] or empty
};

/// `SimBox {bounds_low, bounds_high}` (src/properties/sol_shrinking_box.rs:600-603).
struct SimBox {
    std::array<float, 3> bounds_low{}, bounds_high{};
    std::array<float, 3> extent() const { return {bounds_high[0] - bounds_low[0], bounds_high[1] - bounds_low[1], bounds_high[2] - bounds_low[2]}; }
    std::array<float, 3> center() const { return {0.5f * (bounds_high[0] + bounds_low[0]), 0.5f * (bounds_high[1] + bounds_low[1]), 0.5f * (bounds_high[2] + bounds_low[2])}; }
    /// Writes the cell into a system description and makes it periodic (what `MdState::new` does with `cfg.sim_box`).
    void apply(mdx_system& sys) const {
        for (int k = 0; k < 3; ++k) { sys.box_lo[k] = bounds_low[k]; sys.box_hi[k] = bounds_high[k]; }
        sys.periodic = 1;
    }
};

/// `SimBoxInit::{Pad(f32), Fixed((lo, hi))}` + `SimBoxInit::new_cube(side)` (src/md/mod.rs:656-659; src/ui/panels/md.rs:582-584;
/// src/properties/water_sol.rs:190, crystal.rs:321, logp.rs:119).  `Pad`: the cell is the extent of the atoms plus `pad` on every
/// side - the rule the reference restates itself at src/gromacs/mod.rs:540-576 ("a copy+paste of SimBox::from_atoms in dynamics":
/// bounds_low = min - pad, bounds_high = max + pad).  `new_cube`: a cube of the given edge; where the crate centres it is not
/// visible in the tree - here about `centre` (default: the origin; callers set `recenter_sim_box`, water_sol.rs:192).
struct SimBoxInit {
    enum class Kind { Pad, Fixed } kind = Kind::Pad;
    float pad = 12.f;                         // the UI's default (src/ui/panels/md.rs:582)
    SimBox fixed{};
    static SimBoxInit Pad(float pad_angstrom) { SimBoxInit b; b.kind = Kind::Pad; b.pad = pad_angstrom; return b; }
    static SimBoxInit Fixed(const std::array<float, 3>& lo, const std::array<float, 3>& hi) {
        SimBoxInit b; b.kind = Kind::Fixed; b.fixed.bounds_low = lo; b.fixed.bounds_high = hi; return b;
    }
    static SimBoxInit new_cube(float side, const std::array<float, 3>& centre = {0.f, 0.f, 0.f}) {
        const float h = 0.5f * side;
        return Fixed({centre[0] - h, centre[1] - h, centre[2] - h}, {centre[0] + h, centre[1] + h, centre[2] + h});
    }
    /// `SimBox::from_atoms`: the cell for `n_atoms` positions ([3 n] floats).  Pad with no atoms is an error, as in the reference
    /// (`sim_box_nm` returns None, src/gromacs/mod.rs:571-573).
    SimBox resolve(const float* pos, size_t n_atoms) const {
        if (kind == Kind::Fixed) return fixed;
        if (!pos || n_atoms == 0) throw ParamError("SimBoxInit::Pad needs at least one atom");
        SimBox b;
        for (int k = 0; k < 3; ++k) { b.bounds_low[k] = pos[k]; b.bounds_high[k] = pos[k]; }
        for (size_t i = 1; i < n_atoms; ++i)
            for (int k = 0; k < 3; ++k) {
                b.bounds_low[k] = std::min(b.bounds_low[k], pos[3 * i + k]);
                b.bounds_high[k] = std::max(b.bounds_high[k], pos[3 * i + k]);
            }
        for (int k = 0; k < 3; ++k) { b.bounds_low[k] -= pad; b.bounds_high[k] += pad; }
        return b;
    }
};

class MdState {
public:
    /// `MdState::new(dev, &cfg, &mols, param_set)` (src/md/mod.rs:689).  The handle copies the system.
    static MdState create(const mdx_system& system, const mdx_config& cfg, int device = 0) {
        mdx_handle* h = nullptr;
        check(mdx_create(&system, &cfg, device, &h));
        return MdState(h, system.n_atoms);
    }
    MdState(MdState&& o) noexcept : h_(std::exchange(o.h_, nullptr)), n_(o.n_), n_waters_(o.n_waters_), water_sites_(o.water_sites_) {}
    MdState& operator=(MdState&& o) noexcept {
        if (this != &o) { reset(); h_ = std::exchange(o.h_, nullptr); n_ = o.n_; n_waters_ = o.n_waters_; water_sites_ = o.water_sites_; }
        return *this;
    }
    MdState(const MdState&) = delete;
    MdState& operator=(const MdState&) = delete;
    ~MdState() { reset(); }

    uint32_t n_atoms() const { return n_; }
    mdx_handle* raw() const { return h_; }

    /// `md.step(&dev, dt, external_forces)` repeated `n_steps` times on the device (src/md/mod.rs:716,748;
    /// src/mol_alignment.rs:346).  external_forces: [3N] kcal/mol/Å in caller order, or nullptr.
    void step(float dt, const float* external_forces = nullptr, uint32_t n_steps = 1) {
        check(mdx_step(h_, dt, external_forces, n_steps));
    }
    /// `md.step_count` (src/md/mod.rs:738).
    uint64_t step_count() const { return mdx_step_count(h_); }
    double time_ps() const { return mdx_time_ps(h_); }

    /// per-snapshot `energy_data` of the current state (src/ui/panels/md_viewer.rs:195-257).
    mdx_energies energy() {
        mdx_energies e{};
        check(mdx_energy(h_, &e));
        return e;
    }
    /// `md.atoms[i].posit / .force` read-back (src/mol_alignment.rs:349-352; src/md/mod.rs:843-852).
    std::vector<float> positions() { return download(MDX_POS); }
    std::vector<float> velocities() { return download(MDX_VEL); }
    std::vector<float> forces() { return download(MDX_FORCE); }
    void set_positions(const std::vector<float>& p) { upload(MDX_POS, p); }
    /// The docking loop's pose update (src/docking/mod.rs:81-154): atoms [first, first + p.size()/3) only; the Verlet
    /// list is kept while they stay inside its skin.
    void set_positions_range(uint32_t first, const std::vector<float>& p) {
        check(mdx_upload_range(h_, MDX_POS, first, (uint32_t)(p.size() / 3), p.data()));
    }
    void set_velocities(const std::vector<float>& v) { upload(MDX_VEL, v); }

    /// `md.cell = SimBox::new(lo, hi)` + `md.rebuild_spatial_caches()` (sol_shrinking_box.rs:600-603, 632).
    void set_cell(const SimBox& b) { check(mdx_set_box(h_, b.bounds_low.data(), b.bounds_high.data())); }
    SimBox cell() const {
        SimBox b;
        check(mdx_get_box(h_, b.bounds_low.data(), b.bounds_high.data()));
        return b;
    }
    void rebuild_spatial_caches() { check(mdx_rebuild_spatial_caches(h_)); }
    /// `md.shrink_cell_towards(dev, target_cell, cfg) -> bool` (src/properties/sol_shrinking_box.rs:990).
    bool shrink_cell_towards(const SimBox& target, float box_shrink_per_step) {
        int shrank = 0;
        check(mdx_shrink_cell_towards(h_, target.bounds_low.data(), target.bounds_high.data(), box_shrink_per_step, &shrank));
        return shrank != 0;
    }

    /// `md.minimize_energy(dev, iters, external_forces)` (src/ui/mol_editor.rs:375; src/mol_alignment.rs:356).
    /// Returns the final energies; `iters_done` receives the force evaluations used.
    mdx_energies minimize_energy(uint32_t max_iters, const float* external_forces = nullptr, float f_tol = 0.f,
                                 uint32_t* iters_done = nullptr) {
        mdx_energies e{};
        check(mdx_minimize_energy(h_, max_iters, external_forces, f_tol, &e, iters_done));
        return e;
    }
    /// `md.initialize_velocities(temperature, zero_com_drift)` (sol_shrinking_box.rs:965).
    void initialize_velocities(float temperature, bool zero_com_drift = true, uint64_t seed = 0) {
        check(mdx_initialize_velocities(h_, temperature, zero_com_drift ? 1 : 0, seed));
    }
    /// `Integrator::{VerletVelocity, Leapfrog, LangevinMiddle{gamma}}` (src/ui/panels/md.rs:296-305).
    void set_integrator(int kind, float gamma_per_ps = 1.f, float temperature = 300.f, uint64_t seed = 0) {
        check(mdx_set_integrator(h_, kind, gamma_per_ps, temperature, seed));
    }
    /// `thermostat: Some(tau)` + `temp_target` (src/ui/panels/md.rs:296-305; CSVR per README.md:237-238).
    void set_thermostat(int kind, float temp_target, float tau_ps, uint32_t every_n_steps = 10, uint64_t seed = 0) {
        check(mdx_set_thermostat(h_, kind, temp_target, tau_ps, every_n_steps, seed));
    }
    /// `BarostatCfg{tau, pressure_target}` (src/ui/panels/md.rs:517-557).
    void set_barostat(int kind, float pressure_target_bar = 1.f, float tau_ps = 5.f, float compressibility_per_bar = 0.f,
                      uint32_t every_n_steps = 25) {
        check(mdx_set_barostat(h_, kind, pressure_target_bar, tau_ps, compressibility_per_bar, every_n_steps));
    }
    void set_zero_com_drift(bool on) { check(mdx_set_zero_com_drift(h_, on ? 1 : 0)); }
    /// `md.configure_alchemical_window(dev, mol_index, lambda)` (src/properties/water_sol.rs:556).
    void configure_alchemical_window(uint32_t mol_index, double lambda) {
        check(mdx_configure_alchemical_window(h_, mol_index, lambda));
    }

    /// Soft core of the alchemical window (alpha = 0: linear coupling).
    void set_alchemical_softcore(float alpha = 0.5f, float sigma_min = 3.f) { check(mdx_set_alchemical_softcore(h_, alpha, sigma_min)); }

    /// `md.water` (src/properties/sol_shrinking_box.rs:605-613): where the waters sit in the flat atom array ...
    void set_water_layout(uint32_t first_atom, uint32_t n_waters, uint32_t sites_per_water) {
        check(mdx_set_water_layout(h_, first_atom, n_waters, sites_per_water));
        n_waters_ = n_waters; water_sites_ = sites_per_water;
    }
    /// ... and `md.water[i].{o,h0,h1,m}.posit` / `.force` (which = MDX_POS / MDX_FORCE) as four [3 n_waters] arrays.
    struct Water { std::vector<float> o, h0, h1, m; };
    Water water(int which = MDX_POS) {
        Water w;
        w.o.resize(3 * (size_t)n_waters_); w.h0.resize(w.o.size()); w.h1.resize(w.o.size());
        if (water_sites_ == 4) w.m.resize(w.o.size());
        check(mdx_water_download(h_, which, w.o.data(), w.h0.data(), w.h1.data(), water_sites_ == 4 ? w.m.data() : nullptr));
        return w;
    }
    /// hydrogen bonds in every snapshot's energy data (src/md/viewer.rs:917-960)
    void set_hbond_detection(const std::vector<uint8_t>& is_heavy_nosf, float max_h_acc_dist = 2.5f, float min_angle_deg = 120.f) {
        check(mdx_set_hbond_detection(h_, is_heavy_nosf.empty() ? nullptr : is_heavy_nosf.data(), max_h_acc_dist, min_angle_deg));
    }
    std::vector<mdx_hbond> snapshot_hydrogen_bonds(uint32_t k) {
        std::vector<mdx_hbond> hb(mdx_snapshot_hbond_count(h_, k));
        check(mdx_snapshot_read_hbonds(h_, k, hb.data(), (uint32_t)hb.size()));
        return hb;
    }

    /// One box over several GPUs (the reference is single-device): join a communicator; from then on step /
    /// energy / positions are collective calls.  `id`: the 128 bytes of mdx::comm_unique_id() drawn by rank 0.
    void comm_init(const std::array<uint8_t, MDX_COMM_ID_BYTES>& id, int rank, int world) { check(mdx_comm_init(h_, id.data(), rank, world)); }
    /// ... or, between handles of ONE process (one thread each), an in-process fabric.
    void comm_init_fabric(mdx_fabric* fabric, int rank) { check(mdx_comm_init_fabric(h_, fabric, rank)); }
    /// ... or between processes of one host through POSIX shared memory (verification transport: no RCCL, any devices).
    void comm_init_shm(const std::string& name, int rank, int world) { check(mdx_comm_init_shm(h_, name.c_str(), rank, world)); }
    /// Collective: every transport entry point once on the real wire, results checked.
    void comm_selftest() { check(mdx_comm_selftest(h_)); }

    /// `snapshot_handlers.memory: Some(every_n)` (src/properties/water_sol.rs:185-189).
    void set_snapshot_cadence(uint32_t every_n, bool with_velocities = false) {
        check(mdx_set_snapshot_cadence(h_, every_n, with_velocities ? 1 : 0));
    }
    /// `md.snapshots` cloned out (src/md/mod.rs:121-122).
    std::vector<Snapshot> snapshots(bool with_velocities = false) {
        std::vector<Snapshot> out(mdx_snapshot_count(h_));
        for (uint32_t k = 0; k < out.size(); ++k) {
            Snapshot& s = out[k];
            s.atom_posits.resize(3 * (size_t)n_);
            if (with_velocities) s.atom_velocities.resize(3 * (size_t)n_);
            check(mdx_snapshot_read(h_, k, &s.time_ps, &s.step, &s.energy_data, s.atom_posits.data(),
                                    with_velocities ? s.atom_velocities.data() : nullptr));
        }
        return out;
    }
    /// `md.flush_snapshot_queues()` (src/md/mod.rs:118-120).
    void flush_snapshot_queues() { check(mdx_flush_snapshot_queues(h_)); }

    /// Verlet list in caller atom order as CSR (parity / debugging API).
    std::pair<std::vector<uint32_t>, std::vector<uint32_t>> neighbor_list() {
        std::vector<uint32_t> off((size_t)n_ + 1);
        check(mdx_neighbor_list(h_, off.data(), nullptr));
        std::vector<uint32_t> idx(off[n_]);
        check(mdx_neighbor_list(h_, off.data(), idx.data()));
        return {std::move(off), std::move(idx)};
    }
    mdx_stats stats() {
        mdx_stats s{};
        check(mdx_get_stats(h_, &s));
        return s;
    }
    /// `md.computation_time()` (src/md/mod.rs:740-743): ms spent inside step calls.
    double computation_time() { return stats().wall_ms_sum; }

private:
    MdState(mdx_handle* h, uint32_t n) : h_(h), n_(n) {}
    void reset() { if (h_) { mdx_destroy(h_); h_ = nullptr; } }
    std::vector<float> download(int which) {
        std::vector<float> v(3 * (size_t)n_);
        check(mdx_download(h_, which, v.data()));
        return v;
    }
    void upload(int which, const std::vector<float>& v) {
        if (v.size() != 3 * (size_t)n_) throw ParamError("expected 3N floats");
        check(mdx_upload(h_, which, v.data()));
    }
    mdx_handle* h_ = nullptr;
    uint32_t n_ = 0, n_waters_ = 0, water_sites_ = 0;
};

/// ncclGetUniqueId through the library (rank 0; hand the bytes to the other ranks).
inline std::array<uint8_t, MDX_COMM_ID_BYTES> comm_unique_id() {
    std::array<uint8_t, MDX_COMM_ID_BYTES> id{};
    check(mdx_comm_unique_id(id.data()));
    return id;
}

/// `dynamics::compute_energy_snapshot(dev, &mols, param_set)` (src/md/mod.rs:1036): stateless single point.
inline mdx_energies compute_energy_snapshot(const mdx_system& system, const mdx_config& cfg, int device = 0,
                                            std::vector<float>* forces = nullptr) {
    mdx_energies e{};
    if (forces) forces->resize(3 * (size_t)system.n_atoms);
    check(mdx_single_point(&system, &cfg, device, &e, forces ? forces->data() : nullptr));
    return e;
}

/// Frees the device state the scorer keeps between poses of the same molecules (mdx_single_point_release).
inline void release_single_point_cache() { mdx_single_point_release(); }

/// `run_dynamics_blocking` (src/md/mod.rs:696-724): n steps in one go.
inline void run_dynamics_blocking(MdState& md, uint32_t n_steps, float dt) { md.step(dt, nullptr, n_steps); }

}  // namespace mdx
