// Primal-dual regularisation state machines of the inertia-correction loop — host-side scalars, like in the reference
// (src/Optimization/hiopPDPerturbation.{hpp,cpp}):
//   kind PrimalFirst   hiopPDPerturbationPrimalFirstScalar   hiopPDPerturbation.cpp:108-395
//   kind DualFirst     hiopPDPerturbationDualFirstScalar     :470-626  (normal-equation KKT: dual regularisation is tried first)
//   null_mode          hiopPDPerturbationNull                hiopPDPerturbation.hpp:216-247 (quasi-Newton path: deltas stay 0)
//   randomized         hiop...PrimalFirstRand / DualFirstRand :414-455, :670-711 — the same scalar machines; the VECTORS handed
//                      to the KKT builders are uniform in [min_uniform_ratio, max_uniform_ratio] x scalar (0.9, 1.0,
//                      hiopPDPerturbation.hpp:53-54) instead of constant.  `dirty` records which group of vectors the last call
//                      (re)drew (set_delta_curr_vec(PrimalUpdate | DualUpdate | PDUpdate)); the owner of the device vectors
//                      (kkt_xycyd.hip) consumes and clears it.
// No device code in here: the C-ABI hiopamd_pd_perturbation_* (end of kkt_xycyd.hip) exposes the machines
// on their own (callers that run the correction loop themselves; the CPU test-suite drives them without a GPU).
#pragma once
#include <cmath>

namespace hiopamd {

struct PdPerturb {
  enum Degeneracy { NotEstablished, NotDegenerate, Degenerate };
  enum TestType { NoTest, Dc0Dw0, DcposDw0, Dc0Dwpos, DcposDwpos };
  enum Kind { PrimalFirst = 0, DualFirst = 1 };
  enum Dirty { PrimalUpdate = 1, DualUpdate = 2, PDUpdate = 3 };
  int kind = PrimalFirst;
  bool null_mode = false;
  bool randomized = false;
  double min_uniform_ratio = 0.9, max_uniform_ratio = 1.0;
  int dirty = PDUpdate;
  double wx = 0, wd = 0, cc = 0, cd = 0;
  double wx_last = 0, wd_last = 0, cc_last = 0, cd_last = 0;
  // hiopOptions.cpp:1080-1123 defaults
  double delta_w_min_bar = 1e-20, delta_w_max_bar = 1e20, delta_w_0_bar = 1e-4, kappa_w_minus = 1. / 3,
         kappa_w_plus_bar = 100., kappa_w_plus = 8., delta_c_bar = 1e-8, kappa_c = 0.25;
  double delta_c_min_bar = 1e-20, kappa_c_plus = 10.;   // hiopPDPerturbation.cpp:459-460 (dual-first only)
  Degeneracy hess_degenerate = NotEstablished, jac_degenerate = NotEstablished;
  int num_degen_iters = 0;
  const int num_degen_max_iters = 3;
  TestType test_type = NoTest;
  double mu = 1e-8;

  double compute_delta_c() const { return delta_c_bar * std::pow(mu, kappa_c); }   // :361

  void update_degeneracy_type()   // :108-157
  {
    switch(test_type) {
      case NoTest: return;
      case Dc0Dw0:
        if(hess_degenerate == NotEstablished && jac_degenerate == NotEstablished) {
          hess_degenerate = jac_degenerate = NotDegenerate;
        } else if(hess_degenerate == NotEstablished) {
          hess_degenerate = NotDegenerate;
        } else if(jac_degenerate == NotEstablished) {
          jac_degenerate = NotDegenerate;
        }
        break;
      case DcposDw0:
        if(hess_degenerate == NotEstablished) hess_degenerate = NotDegenerate;
        if(jac_degenerate == NotEstablished) {
          if(++num_degen_iters >= num_degen_max_iters) jac_degenerate = Degenerate;
        }
        break;
      case Dc0Dwpos:
        if(jac_degenerate == NotEstablished) jac_degenerate = NotDegenerate;
        if(hess_degenerate == NotEstablished) {
          if(++num_degen_iters >= num_degen_max_iters) hess_degenerate = Degenerate;
        }
        break;
      case DcposDwpos:
        if(++num_degen_iters >= num_degen_max_iters) hess_degenerate = jac_degenerate = Degenerate;
        break;
    }
  }

  void save_last()   // :168-179, :474-485
  {
    if(wx > 0.) wx_last = wx;
    if(wd > 0.) wd_last = wd;
    if(cc > 0.) cc_last = cc;
    if(cd > 0.) cd_last = cd;
  }

  // ---- primal first -------------------------------------------------------------------------------------
  bool guts_wrong_inertia()   // :331-358
  {
    if(wx == 0.) {
      wx = (wx_last == 0.) ? delta_w_0_bar : std::fmax(delta_w_min_bar, wx_last * kappa_w_minus);
    } else {
      wx = (wx_last == 0. || 1e5 * wx_last < wx) ? kappa_w_plus_bar * wx : kappa_w_plus * wx;
    }
    wd = wx;
    dirty |= PrimalUpdate;   // :349
    if(wx > delta_w_max_bar) {
      wx_last = wd_last = 0.;
      return false;
    }
    return true;
  }

  bool pf_compute_initial_deltas()   // :161-212
  {
    double delta_temp = 0.0, delta_temp2 = 0.0;
    update_degeneracy_type();
    save_last();
    test_type = (hess_degenerate == NotEstablished || jac_degenerate == NotEstablished) ? Dc0Dw0 : NoTest;
    delta_temp = (jac_degenerate == Degenerate) ? compute_delta_c() : 0.0;
    cc = cd = delta_temp;
    if(hess_degenerate == Degenerate) {
      wx = wd = 0.;
      if(!guts_wrong_inertia()) return false;
      // the reference then assigns its two locals, which the call above never writes (:203-209)
    } else {
      delta_temp = delta_temp2 = 0.;
    }
    wx = delta_temp;
    wd = delta_temp2;
    dirty = PDUpdate;   // :210
    return true;
  }

  bool pf_compute_perturb_wrong_inertia()   // :215-243
  {
    update_degeneracy_type();
    bool ret = guts_wrong_inertia();
    if(!ret && cc == 0.) {
      wx = wd = 0.;
      cc = cd = compute_delta_c();
      test_type = NoTest;
      if(hess_degenerate == Degenerate) hess_degenerate = NotEstablished;
      ret = guts_wrong_inertia();
      dirty = PDUpdate;   // :236
    } else {
      dirty |= PrimalUpdate;   // :238
    }
    return ret;
  }

  bool pf_compute_perturb_singularity()   // :248-325
  {
    bool bret = true;
    if(hess_degenerate == NotEstablished || jac_degenerate == NotEstablished) {
      switch(test_type) {
        case Dc0Dw0:
          if(jac_degenerate == NotEstablished) {
            cc = cd = compute_delta_c();
            test_type = DcposDw0;
          } else {
            if(!guts_wrong_inertia()) {
              bret = false;
              break;
            }
            test_type = Dc0Dwpos;
          }
          break;
        case DcposDw0:
          cd = cc = 0.;
          if(!guts_wrong_inertia()) {
            bret = false;
            break;
          }
          test_type = Dc0Dwpos;
          break;
        case Dc0Dwpos:
          cc = cd = compute_delta_c();
          if(!guts_wrong_inertia()) {
            bret = false;
            break;
          }
          test_type = DcposDwpos;
          break;
        case DcposDwpos:
          if(!guts_wrong_inertia()) bret = false;
          break;
        case NoTest: bret = false; break;   // the reference asserts here (:302)
      }
    } else {
      if(cc > 0.) {
        if(!guts_wrong_inertia()) bret = false;
      } else {
        cd = cc = compute_delta_c();
      }
    }
    dirty = PDUpdate;   // :319
    return bret;
  }

  // ---- dual first ---------------------------------------------------------------------------------------
  bool compute_dual_perturb_impl()   // :558-589
  {
    if(cc == 0.) {
      cc = (cc_last == 0.) ? std::fmax(delta_c_min_bar, delta_c_bar * std::pow(mu, kappa_c))
                           : std::fmax(delta_c_min_bar, cc_last * kappa_w_minus);
    } else {
      cc = (cc_last == 0. || 1e5 * cc_last < cc) ? kappa_w_plus_bar * cc : kappa_c_plus * cc;
    }
    cd = cc;
    dirty |= DualUpdate;   // :578
    if(cc > delta_w_max_bar) {
      cc_last = cd_last = 0.;
      return false;
    }
    return true;
  }

  bool compute_primal_perturb_impl()   // :591-620
  {
    if(wx == 0.) {
      wx = (wx_last == 0.) ? delta_w_0_bar : std::fmax(delta_w_min_bar, wx_last * kappa_w_minus);
    } else {
      wx = (wx_last == 0. || 1e5 * wx_last < wx) ? kappa_w_plus_bar * wx : kappa_w_plus * wx;
    }
    wd = wx;
    dirty |= PrimalUpdate;   // :610
    if(wx > delta_w_max_bar) {
      wx_last = wd_last = 0.;
      return false;
    }
    return true;
  }

  bool df_compute_initial_deltas()   // :470-512
  {
    update_degeneracy_type();
    save_last();
    test_type = (hess_degenerate == NotEstablished || jac_degenerate == NotEstablished) ? Dc0Dw0 : NoTest;
    cc = cd = 0.;
    if(jac_degenerate == Degenerate) {
      if(!compute_dual_perturb_impl()) return false;
    }
    wx = wd = 0.;
    if(hess_degenerate == Degenerate) {
      if(!compute_primal_perturb_impl()) return false;
    }
    dirty = PDUpdate;   // :507
    return true;
  }

  bool df_compute_perturb_wrong_inertia()   // :514-547
  {
    update_degeneracy_type();
    bool ret = compute_dual_perturb_impl();
    if(!ret && wx == 0.) {
      cc = cd = 0.;
      ret = compute_primal_perturb_impl();
      if(!ret) return ret;
      test_type = NoTest;
      if(jac_degenerate == Degenerate) jac_degenerate = NotEstablished;
      ret = compute_dual_perturb_impl();
      dirty |= PrimalUpdate;   // :539
    }
    dirty |= DualUpdate;   // :542
    return ret;
  }

  // ---- dispatch (the virtual interface of hiopPDPerturbation) ------------------------------------------
  bool compute_initial_deltas()
  {
    if(null_mode) return true;
    return kind == DualFirst ? df_compute_initial_deltas() : pf_compute_initial_deltas();
  }
  bool compute_perturb_wrong_inertia()
  {
    if(null_mode) return true;
    return kind == DualFirst ? df_compute_perturb_wrong_inertia() : pf_compute_perturb_wrong_inertia();
  }
  bool compute_perturb_singularity()
  {
    if(null_mode) return true;
    return kind == DualFirst ? df_compute_perturb_wrong_inertia() /* :549-556 */ : pf_compute_perturb_singularity();
  }
};

}  // namespace hiopamd
