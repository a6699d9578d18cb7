// mi355_arma.hpp -- Armadillo-facing C++ operator signatures over the C ABI
// (include/mi355_interp.h): arma::vec / arma::mat in, interpolated arma::vec
// out, as BASELINE.json's north_star asks.  Header-only; link libmi355interp.so.
//
//   mi355::interp1(X, Y, XI, YI)            == arma::interp1(X, Y, XI, YI, "linear")
//   mi355::Interp1Table tab(X, Y); tab(XI, YI)   table resident in HBM across calls
//   mi355::interp2(X, Y, Z, XI, YI, ZI)     scattered bilinear, Z = arma::mat(Y.n_elem, X.n_elem)
//   mi355::restrict_to_horizon(...)         RestrictKernel (EventDrivenMap.cu:769-785) on host vectors
//   mi355::DeviceGroup grp(8); mi355::GroupInterp1Table tab(grp, X, Y); tab(XI, YI)
//                                           the same call with the queries sharded over the GPUs of the node
//
// Error convention: the reference aborts on any device error (CUDA_CALL ->
// fprintf(stderr) + exit(-1), EventDrivenMap.cu:18-54) and on bad arguments
// (assert); Armadillo itself throws.  These wrappers throw std::runtime_error
// carrying mi_last_error(); mi355::abort_on_error(true) restores print + exit(-1).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <cstdint>
#include <string>
#include <vector>

#include "mi355_arma_compat.hpp"
#include "mi355_interp.h"

namespace mi355 {

inline bool& abort_flag() { static bool f = false; return f; }
inline void abort_on_error(bool on) { abort_flag() = on; }

inline void check(mi_status st, const mi_ctx* ctx, const char* where)
{
    if (st == MI_OK) return;
    std::string msg = std::string(where) + ": " + mi_last_error(ctx);
    if (abort_flag()) {
        std::fprintf(stderr, "%s\n", msg.c_str());
        std::exit(-1);
    }
    throw std::runtime_error(msg);
}

// One context per device, created on first use.
class Device {
  public:
    explicit Device(int ordinal = 0) : ctx_(nullptr) { check(mi_ctx_create(ordinal, &ctx_), nullptr, "mi_ctx_create"); }
    ~Device() { mi_ctx_destroy(ctx_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    mi_ctx* get() const { return ctx_; }
    static Device& instance()
    {
        static Device d(0);
        return d;
    }

  private:
    mi_ctx* ctx_;
};

// A 1-D table kept in HBM: build once, interpolate many query vectors.
class Interp1Table {
  public:
    // sanitise = true reproduces arma::interp1's default (X sorted and de-duplicated first);
    // false is its "*linear" fast path (X must already be strictly increasing).
    Interp1Table(const arma::vec& X, const arma::vec& Y, bool sanitise = true, Device& dev = Device::instance())
        : dev_(dev), g_(nullptr)
    {
        if (X.n_elem != Y.n_elem) throw std::invalid_argument("interp1(): X and Y must have the same number of elements");
        check(mi_grid1_create(dev_.get(), X.memptr(), Y.memptr(), X.n_elem, sanitise ? MI_GRID_SANITISE : 0u, &g_),
              dev_.get(), "mi_grid1_create");
    }
    ~Interp1Table() { mi_grid1_destroy(g_); }
    Interp1Table(const Interp1Table&) = delete;
    Interp1Table& operator=(const Interp1Table&) = delete;

    void operator()(const arma::vec& XI, arma::vec& YI, double extrap_val = std::numeric_limits<double>::quiet_NaN()) const
    {
        YI.set_size(XI.n_elem);
        check(mi_interp1_f64_host(dev_.get(), g_, XI.memptr(), YI.memptr(), XI.n_elem, extrap_val), dev_.get(),
              "mi_interp1_f64_host");
    }

  private:
    Device& dev_;
    mi_grid1* g_;
};

// arma::interp1(X, Y, XI, YI, "linear", extrap_val)
inline void interp1(const arma::vec& X, const arma::vec& Y, const arma::vec& XI, arma::vec& YI,
                    double extrap_val = std::numeric_limits<double>::quiet_NaN(), Device& dev = Device::instance())
{
    if (X.n_elem != Y.n_elem) throw std::invalid_argument("interp1(): X and Y must have the same number of elements");
    YI.set_size(XI.n_elem);
    check(mi_interp1_f64(dev.get(), X.memptr(), Y.memptr(), X.n_elem, XI.memptr(), YI.memptr(), XI.n_elem, extrap_val),
          dev.get(), "mi_interp1_f64");
}

// Scattered bilinear interpolation: ZI[k] = Z(YI[k], XI[k]); Z is Y.n_elem x X.n_elem (rows follow Y), the
// layout arma::interp2 uses for its Z argument.  One result per query PAIR (not arma::interp2's gridded output).
inline void interp2(const arma::vec& X, const arma::vec& Y, const arma::mat& Z, const arma::vec& XI,
                    const arma::vec& YI, arma::vec& ZI,
                    double extrap_val = std::numeric_limits<double>::quiet_NaN(), Device& dev = Device::instance())
{
    if (Z.n_rows != Y.n_elem || Z.n_cols != X.n_elem) throw std::invalid_argument("interp2(): Z must be Y.n_elem x X.n_elem");
    if (XI.n_elem != YI.n_elem) throw std::invalid_argument("interp2(): XI and YI must have the same number of elements");
    mi_grid2* g = nullptr;
    check(mi_grid2_create(dev.get(), X.memptr(), X.n_elem, Y.memptr(), Y.n_elem, Z.memptr(), 0u, &g), dev.get(),
          "mi_grid2_create");
    ZI.set_size(XI.n_elem);
    mi_status st = mi_interp2_f64_host(dev.get(), g, XI.memptr(), YI.memptr(), ZI.memptr(), XI.n_elem, extrap_val);
    mi_grid2_destroy(g);
    check(st, dev.get(), "mi_interp2_f64_host");
}

// RestrictKernel (EventDrivenMap.cu:769-785) on host vectors: position at t = final_time from the last event
// before (t0, i0) and the first event after (t1, i1) the horizon, on the implicit grid x_i = -L + 2L/N * i.
inline void restrict_to_horizon(const arma::fvec& t0, const std::vector<uint16_t>& i0, const arma::fvec& t1,
                                const std::vector<uint16_t>& i1, float final_time, float half_length,
                                unsigned int n_grid, arma::fvec& out, Device& dev = Device::instance())
{
    const size_t n = t0.n_elem;
    if (t1.n_elem != n || i0.size() != n || i1.size() != n) throw std::invalid_argument("restrict_to_horizon(): size mismatch");
    out.set_size(n);
    check(mi_restrict_f32_host(dev.get(), t0.memptr(), i0.data(), t1.memptr(), i1.data(), final_time, half_length, n_grid,
                               out.memptr(), n),
          dev.get(), "mi_restrict_f32_host");
}

// Several GPUs of one node behind the same call shapes (mi_group_*): contiguous query shards, the table
// replicated, one call per interpolation.  devices = {0, 1, ..}; a repeated ordinal rehearses the sharding on one GPU.
class DeviceGroup {
  public:
    explicit DeviceGroup(int ndev) : g_(nullptr) { check(mi_group_create(ndev, nullptr, &g_), nullptr, "mi_group_create"); }
    explicit DeviceGroup(const std::vector<int>& devices) : g_(nullptr)
    {
        check(mi_group_create((int)devices.size(), devices.data(), &g_), nullptr, "mi_group_create");
    }
    ~DeviceGroup() { mi_group_destroy(g_); }
    DeviceGroup(const DeviceGroup&) = delete;
    DeviceGroup& operator=(const DeviceGroup&) = delete;
    mi_group* get() const { return g_; }
    int size() const { return mi_group_size(g_); }

  private:
    mi_group* g_;
};

class GroupInterp1Table {
  public:
    GroupInterp1Table(DeviceGroup& grp, const arma::vec& X, const arma::vec& Y, bool sanitise = true) : grp_(grp), t_(nullptr)
    {
        if (X.n_elem != Y.n_elem) throw std::invalid_argument("interp1(): X and Y must have the same number of elements");
        check(mi_group_grid1_create(grp_.get(), X.memptr(), Y.memptr(), X.n_elem, sanitise ? MI_GRID_SANITISE : 0u, &t_), nullptr,
              "mi_group_grid1_create");
    }
    ~GroupInterp1Table() { mi_group_grid1_destroy(t_); }
    GroupInterp1Table(const GroupInterp1Table&) = delete;
    GroupInterp1Table& operator=(const GroupInterp1Table&) = delete;
    void operator()(const arma::vec& XI, arma::vec& YI, double extrap_val = std::numeric_limits<double>::quiet_NaN()) const
    {
        YI.set_size(XI.n_elem);
        check(mi_group_interp1_f64_host(grp_.get(), t_, XI.memptr(), YI.memptr(), XI.n_elem, extrap_val), nullptr,
              "mi_group_interp1_f64_host");
    }

  private:
    DeviceGroup& grp_;
    mi_group_grid1* t_;
};

// Scattered bilinear interpolation (BASELINE config 3) over the group: Z = arma::mat(Y.n_elem, X.n_elem) replicated, the
// query pairs sharded; one result per pair, as mi355::interp2.
class GroupInterp2Table {
  public:
    GroupInterp2Table(DeviceGroup& grp, const arma::vec& X, const arma::vec& Y, const arma::mat& Z) : grp_(grp), t_(nullptr)
    {
        if (Z.n_rows != Y.n_elem || Z.n_cols != X.n_elem) throw std::invalid_argument("interp2(): Z must be Y.n_elem x X.n_elem");
        check(mi_group_grid2_create(grp_.get(), X.memptr(), X.n_elem, Y.memptr(), Y.n_elem, Z.memptr(), 0u, &t_), nullptr,
              "mi_group_grid2_create");
    }
    ~GroupInterp2Table() { mi_group_grid2_destroy(t_); }
    GroupInterp2Table(const GroupInterp2Table&) = delete;
    GroupInterp2Table& operator=(const GroupInterp2Table&) = delete;
    void operator()(const arma::vec& XI, const arma::vec& YI, arma::vec& ZI,
                    double extrap_val = std::numeric_limits<double>::quiet_NaN()) const
    {
        if (XI.n_elem != YI.n_elem) throw std::invalid_argument("interp2(): XI and YI must have the same number of elements");
        ZI.set_size(XI.n_elem);
        check(mi_group_interp2_f64_host(grp_.get(), t_, XI.memptr(), YI.memptr(), ZI.memptr(), XI.n_elem, extrap_val), nullptr,
              "mi_group_interp2_f64_host");
    }

  private:
    DeviceGroup& grp_;
    mi_group_grid2* t_;
};

}  // namespace mi355
