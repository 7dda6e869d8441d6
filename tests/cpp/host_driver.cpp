// host_driver.cpp — the compiled-language host above the C ABI (include/mdx.hpp), exercising the surface the
// reference's Rust host calls: MdState::new / step / minimize_energy / initialize_velocities / snapshots /
// compute_energy_snapshot and the ParamError path (/root/reference src/md/mod.rs:689-750, 1036;
// src/properties/sol_shrinking_box.rs:962-995).  Built and run by tests/test_cpp_host.py; exit code 0 = all checks hold.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "mdx.hpp"

namespace {
struct WaterBox {   // flexible 3-site TIP3P on a lattice, orientations from a small LCG (standard Amber values)
    std::vector<float> pos, vel, mass, charge, bond_k, bond_r0, angle_k, angle_t0;
    std::vector<uint32_t> lj_type, bond_idx, angle_idx, excl_off, excl_idx, mol_start;
    float sigma[2] = {3.15061f, 0.f}, eps[2] = {0.1521f, 0.f};
    mdx_system sys{};
    explicit WaterBox(int n) {
        const float sp = 3.1034f, roh = 0.9572f, th = 104.52f * 3.14159265f / 180.f;
        uint64_t st = 12345;
        auto rnd = [&] { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xFFFFFF) / 16777216.f; };
        for (int ix = 0; ix < n; ++ix) for (int iy = 0; iy < n; ++iy) for (int iz = 0; iz < n; ++iz) {
            const uint32_t o = (uint32_t)pos.size() / 3;
            const float c[3] = {(ix + 0.5f) * sp, (iy + 0.5f) * sp, (iz + 0.5f) * sp};
            const float a = 6.2831853f * rnd(), b = std::acos(2.f * rnd() - 1.f), g = 6.2831853f * rnd();
            // two unit vectors spanning the molecular plane
            const float u[3] = {std::sin(b) * std::cos(a), std::sin(b) * std::sin(a), std::cos(b)};
            float w[3] = {std::cos(b) * std::cos(a), std::cos(b) * std::sin(a), -std::sin(b)};
            const float x[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
            for (int k = 0; k < 3; ++k) w[k] = std::cos(g) * w[k] + std::sin(g) * x[k];
            for (int k = 0; k < 3; ++k) pos.push_back(c[k]);
            for (int s = -1; s <= 1; s += 2)
                for (int k = 0; k < 3; ++k) pos.push_back(c[k] + roh * (std::cos(th / 2) * u[k] + s * std::sin(th / 2) * w[k]));
            const float m[3] = {15.9994f, 1.008f, 1.008f}, q[3] = {-0.834f, 0.417f, 0.417f};
            for (int k = 0; k < 3; ++k) { mass.push_back(m[k]); charge.push_back(q[k]); lj_type.push_back(k ? 1u : 0u); }
            for (uint32_t hh = 1; hh <= 2; ++hh) { bond_idx.push_back(o); bond_idx.push_back(o + hh); bond_k.push_back(553.f); bond_r0.push_back(roh); }
            angle_idx.insert(angle_idx.end(), {o + 1, o, o + 2}); angle_k.push_back(100.f); angle_t0.push_back(th);
            mol_start.push_back(o);
        }
        const uint32_t N = (uint32_t)mass.size();
        excl_off.push_back(0);
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t o = i - i % 3;
            for (uint32_t j = o; j < o + 3; ++j) if (j != i) excl_idx.push_back(j);
            excl_off.push_back((uint32_t)excl_idx.size());
        }
        vel.assign(3 * (size_t)N, 0.f);
        sys.n_atoms = N; sys.pos = pos.data(); sys.vel = vel.data(); sys.mass = mass.data(); sys.charge = charge.data();
        sys.lj_type = lj_type.data(); sys.n_lj_types = 2; sys.lj_sigma = sigma; sys.lj_eps = eps;
        sys.n_bonds = (uint32_t)bond_k.size(); sys.bond_idx = bond_idx.data(); sys.bond_k = bond_k.data(); sys.bond_r0 = bond_r0.data();
        sys.n_angles = (uint32_t)angle_k.size(); sys.angle_idx = angle_idx.data(); sys.angle_k = angle_k.data(); sys.angle_theta0 = angle_t0.data();
        sys.excl_offsets = excl_off.data(); sys.excl_idx = excl_idx.data();
        sys.n_mols = (uint32_t)mol_start.size(); sys.mol_start = mol_start.data();
        sys.periodic = 1;
        for (int k = 0; k < 3; ++k) { sys.box_lo[k] = 0.f; sys.box_hi[k] = n * sp; }
    }
};

int fails = 0;
void expect(bool ok, const char* what) {
    std::printf("%s  %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) ++fails;
}
}  // namespace

int main() {
    if (mdx::device_count() < 1) { std::puts("no gfx950 device: the library has no CPU path"); return 77; }
    WaterBox box(10);                                    // 3000 atoms, 31 A cube
    mdx_config cfg = mdx::default_config();
    cfg.lj_cutoff = 9.f; cfg.coulomb_cutoff = 9.f; cfg.skin = 1.5f; cfg.coulomb_mode = MDX_COULOMB_REACTION;

    // ParamError { descrip } for a system the engine cannot build
    try {
        mdx_system bad = box.sys; bad.n_atoms = 0;
        (void)mdx::MdState::create(bad, cfg);
        expect(false, "ParamError for an empty system");
    } catch (const mdx::ParamError& e) { expect(std::string(e.what()).size() > 0, "ParamError carries a description"); }

    mdx::MdState md = mdx::MdState::create(box.sys, cfg);
    const mdx_energies e0 = md.energy();
    const mdx_energies em = md.minimize_energy(60);
    expect(em.potential < e0.potential, "minimize_energy lowers the potential energy");

    md.initialize_velocities(300.f, true, 7);
    const mdx_energies et = md.energy();
    expect(std::fabs(et.temperature - 300.0) < 30.0, "initialize_velocities(300 K) gives ~300 K");

    // F = -dE/dx by central differences through the stateless scorer, at the minimised geometry
    {
        std::vector<float> x = md.positions(), f = md.forces();
        std::vector<float> saved = box.pos;
        double worst = 0.0;
        for (uint32_t coord : {0u, 4u, 3u * 777u + 2u}) {
            const float h = 0.01f;
            box.pos = x; box.sys.pos = box.pos.data();
            box.pos[coord] = x[coord] + h; const double ep = mdx::compute_energy_snapshot(box.sys, cfg).potential;
            box.pos[coord] = x[coord] - h; const double en = mdx::compute_energy_snapshot(box.sys, cfg).potential;
            const double fd = -(ep - en) / (2.0 * h);
            worst = std::fmax(worst, std::fabs(fd - f[coord]) / std::fmax(1.0, std::fabs(fd)));
        }
        box.pos = saved; box.sys.pos = box.pos.data();
        expect(worst < 0.05, "forces are minus the gradient of compute_energy_snapshot");
    }

    // NVE: 10-step bursts like the GUI (src/md/mod.rs:737), snapshots every 50 steps, total energy conserved
    md.set_snapshot_cadence(50, true);
    const mdx_energies ea = md.energy();
    for (int burst = 0; burst < 20; ++burst) md.step(0.0005f, nullptr, 10);
    const mdx_energies eb = md.energy();
    expect(md.step_count() == 200, "step_count advances by the steps taken");
    const double drift = std::fabs((eb.potential + eb.kinetic) - (ea.potential + ea.kinetic));
    expect(drift < 0.02 * ea.kinetic, "NVE total energy is conserved over 200 steps");
    const auto snaps = md.snapshots(true);
    expect(snaps.size() == 4 && snaps.back().step == 200 && snaps[0].atom_velocities.size() == 3u * md.n_atoms(),
           "memory snapshots at the cadence, with velocities");
    md.flush_snapshot_queues();
    expect(md.snapshots().empty(), "flush_snapshot_queues empties the queue");

    // the dual pair list ran and changed nothing a fresh evaluation can see
    const mdx_stats st = md.stats();
    expect(st.prune_passes > 0 && st.n_inner_cluster_pairs < st.n_cluster_pairs, "dual pair list active in the step loop");

    mdx::MdState moved = std::move(md);                   // ownership moves like the Rust value
    mdx::run_dynamics_blocking(moved, 5, 0.0005f);
    expect(moved.step_count() == 205 && md.raw() == nullptr, "MdState is move-only; run_dynamics_blocking steps it");

    // md.water views and a ligand-sized pose update on the moved handle
    {
        moved.set_water_layout(0, box.sys.n_atoms / 3, 3);
        const auto w = moved.water(MDX_POS);
        const auto x = moved.positions();
        expect(w.o.size() == box.sys.n_atoms && w.m.empty(), "md.water views have one row per water (3-site: no M)");
        // the oxygen is the flat array's; the hydrogens are the flat array's atoms in the periodic image next to their oxygen
        // (the engine wraps every atom on its own; the views hand molecules out whole)
        bool same = true, whole = true;
        for (uint32_t i = 0; i < box.sys.n_atoms / 3 && same; ++i)
            for (int k = 0; k < 3; ++k) {
                const float L = box.sys.box_hi[k] - box.sys.box_lo[k];
                const float d0 = w.h0[3 * i + k] - x[9 * i + 3 + k], d1 = w.h1[3 * i + k] - x[9 * i + 6 + k];
                same = same && w.o[3 * i + k] == x[9 * i + k] && std::fabs(d0 - L * std::round(d0 / L)) < 1e-4f && std::fabs(d1 - L * std::round(d1 / L)) < 1e-4f;
                whole = whole && std::fabs(w.h0[3 * i + k] - w.o[3 * i + k]) < 1.3f && std::fabs(w.h1[3 * i + k] - w.o[3 * i + k]) < 1.3f;
            }
        expect(same && whole, "md.water[i].{o,h0,h1}.posit are views over the flat atom array, molecules whole");
        const uint64_t rb = moved.stats().rebuild_count;
        std::vector<float> lig(x.begin() + 30, x.begin() + 60);
        for (float& v : lig) v += 0.05f;
        moved.set_positions_range(10, lig);
        (void)moved.energy();
        expect(moved.stats().rebuild_count == rb, "a small pose update (mdx_upload_range) keeps the Verlet list");
    }

    // One box decomposed over two ranks of THIS process (std::thread + the in-process fabric): the whole multi-GPU path
    // - partition, halo exchange, stale-list protocol, energy reduction - runs below the ABI, no Python anywhere.
    {
        WaterBox big(14);                                 // 8232 atoms, 43.4 A cube: wide enough to cut in two
        mdx::MdState ref = mdx::MdState::create(big.sys, cfg);
        const mdx_energies r0 = ref.energy();
        ref.step(0.0005f, nullptr, 30);
        const std::vector<float> xr = ref.positions();
        mdx_fabric* fabric = mdx_fabric_create(2);
        std::vector<float> xd[2]; mdx_energies d0[2]; int bad[2] = {0, 0};
        auto rank_main = [&](int rank) {
            try {
                mdx::MdState md2 = mdx::MdState::create(big.sys, cfg);
                md2.comm_init_fabric(fabric, rank);
                d0[rank] = md2.energy();
                md2.step(0.0005f, nullptr, 30);
                xd[rank] = md2.positions();
            } catch (const std::exception& e) { std::printf("rank %d: %s\n", rank, e.what()); bad[rank] = 1; mdx_fabric_abort(fabric); }
        };
        std::thread t0(rank_main, 0), t1(rank_main, 1);
        t0.join(); t1.join();
        mdx_fabric_destroy(fabric);
        expect(!bad[0] && !bad[1], "two decomposed ranks (std::thread + fabric) ran");
        if (!bad[0] && !bad[1]) {
            expect(std::fabs(d0[0].potential - r0.potential) < 1e-5 * std::fabs(r0.potential) + 0.05 && d0[0].potential == d0[1].potential,
                   "decomposed energies are the totals of the box, identical on both ranks");
            double s2 = 0.0; const double L = big.sys.box_hi[0];
            for (size_t k = 0; k < xr.size(); ++k) { double d = xd[0][k] - xr[k]; d -= std::round(d / L) * L; s2 += d * d; }
            expect(std::sqrt(s2 / (xr.size() / 3)) < 2e-3 && xd[0] == xd[1], "decomposed 30-step trajectory follows the single-handle one");
        }
    }

    std::printf("%s\n", fails ? "FAILED" : "ALL OK");
    return fails ? 1 : 0;
}
