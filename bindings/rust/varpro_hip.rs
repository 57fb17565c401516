//! Reference-side shim over `libvarpro_hip.so` (SOURCE ONLY: there is no rustc/cargo in the build image, so this
//! file has never been compiled; it is the binding a varpro maintainer would add).
//!
//! It implements the reference's own traits on top of the C ABI of `include/varpro_hip.h`:
//!   * `levenberg_marquardt::LeastSquaresProblem` for `GpuSeparableProblem`
//!     (replaces `impl LeastSquaresProblem for SeparableProblem`, src/solvers/levmar/mod.rs:22-202)
//!   * `varpro::model::SeparableNonlinearModel` for `GpuModel` (src/model/mod.rs:239-363)
//! so that the unchanged `LevMarSolver::fit` (src/solvers/levmar/mod.rs:238-254) drives the GPU kernels, and adds
//! `fit_batch`, the device-resident batched fit (`vp_fit`).
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_void, CStr};
use std::ptr::null_mut;

use levenberg_marquardt::LeastSquaresProblem;
use nalgebra::storage::Owned;
use nalgebra::{DMatrix, DVector, Dyn};
use varpro::model::SeparableNonlinearModel;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct vp_model_desc {
    pub n_basis: i32,
    pub n_params: i32,
    pub kind: [i32; 8],
    pub param: [[i32; 2]; 8],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct vp_lm_opts {
    pub ftol: f64,
    pub xtol: f64,
    pub gtol: f64,
    pub stepbound: f64,
    pub patience: i32,
    pub scale_diag: i32,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct vp_report {
    pub termination: i32,
    pub n_evals: i32,
    pub objective: f64,
}
pub enum vp_batch {}

pub const VP_BASIS_CONST: i32 = 0;
pub const VP_BASIS_EXP_DECAY: i32 = 1;
pub const VP_BASIS_EXP_RATE: i32 = 2;
pub const VP_BASIS_EXP_COS: i32 = 3;
pub const VP_BASIS_SIN_PHASE: i32 = 4;
pub const VP_F64: i32 = 0;
pub const VP_FLAG_OWN_STREAM: i32 = 8;
pub const VP_FLAG_NO_GRID_RECURRENCE: i32 = 16;

#[link(name = "varpro_hip")]
extern "C" {
    pub fn vp_batch_create(h: *mut *mut vp_batch, model: *const vp_model_desc, dtype: i32, m: i64, s: i64, b: i64,
        t: *const c_void, y: *const c_void, w: *const c_void, svd_epsilon: f64, flags: i32, device: i32,
        hip_stream: *mut c_void) -> i32;
    pub fn vp_batch_destroy(h: *mut vp_batch);
    pub fn vp_set_params(h: *mut vp_batch, alpha: *const c_void) -> i32;
    pub fn vp_params(h: *mut vp_batch, alpha_out: *mut c_void) -> i32;
    pub fn vp_residuals(h: *mut vp_batch, r_out: *mut c_void, status: *mut i32) -> i32;
    pub fn vp_jacobian(h: *mut vp_batch, j_out: *mut c_void, status: *mut i32) -> i32;
    pub fn vp_linear_coeffs(h: *mut vp_batch, c_out: *mut c_void, status: *mut i32) -> i32;
    pub fn vp_weighted_data(h: *mut vp_batch, yw_out: *mut c_void) -> i32;
    pub fn vp_set_observations(h: *mut vp_batch, y: *const c_void) -> i32;
    pub fn vp_cost(h: *mut vp_batch, cost_out: *mut f64) -> i32;
    pub fn vp_evaluate(h: *mut vp_batch, alpha: *const c_void, r: *mut c_void, j: *mut c_void, c: *mut c_void,
        cost: *mut f64, status: *mut i32) -> i32;
    pub fn vp_basis(h: *mut vp_batch, alpha: *const c_void, phi: *mut c_void, dphi: *mut c_void, flags: i32) -> i32;
    pub fn vp_lm_opts_default(o: *mut vp_lm_opts, dtype: i32);
    pub fn vp_fit(h: *mut vp_batch, opts: *const vp_lm_opts, alpha_inout: *mut c_void, c_out: *mut c_void,
        rep: *mut vp_report) -> i32;
    pub fn vp_best_fit(h: *mut vp_batch, fit_out: *mut c_void) -> i32;
    /// parity diagnostics of the fp64-Gram fit kernel: {1/2||r||^2, c, J^T r, J^T J} at alpha, per problem
    pub fn vp_debug_gram_evaluate(h: *mut vp_batch, alpha: *const c_void, out: *mut f64) -> i32;
    pub fn vp_summary(h: *mut vp_batch, out: *mut f64) -> i32;
    pub fn vp_summary_device(h: *mut vp_batch, dev_out4: *mut f64) -> i32;
    pub fn vp_global_fit_condition(h: *mut vp_batch, cond_out: *mut f64) -> i32;
    pub fn vp_fit_trace(h: *mut vp_batch, opts: *const vp_lm_opts, alpha_inout: *mut c_void, c_out: *mut c_void,
        rep: *mut vp_report, trace_out: *mut f64, trace_rows: i32) -> i32;
    pub fn vp_statistics(h: *mut vp_batch, cov_out: *mut c_void, chi2_out: *mut f64, sigma_out: *mut c_void,
        status: *mut i32) -> i32;
    pub fn vp_set_rhs_allreduce(h: *mut vp_batch,
        f: Option<extern "C" fn(*mut c_void, i64, *mut c_void, *mut c_void) -> i32>, user: *mut c_void,
        global_rhs_count: i64) -> i32;
    /// single-RHS fit kernel selection: 0 = automatic, 1 = one wavefront per problem, 2 = persistent slot kernel
    pub fn vp_set_fit_kernel(h: *mut vp_batch, which: i32) -> i32;
    pub fn vp_set_timing(h: *mut vp_batch, enable: i32) -> i32;
    pub fn vp_last_kernel_ms(h: *mut vp_batch, which: i32, ms: *mut f32) -> i32;
    pub fn vp_synchronize(h: *mut vp_batch) -> i32;
    pub fn vp_last_error_detail() -> i32;
    pub fn vp_version() -> *const c_char;
    pub fn vp_device_count() -> i32;
    pub fn vp_last_error() -> *const c_char;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(vp_last_error()).to_string_lossy().into_owned() }
}

/// One separable problem (S right-hand sides) whose state lives on the GPU.
pub struct GpuSeparableProblem {
    h: *mut vp_batch,
    m: usize,
    s: usize,
    n: usize,
    q: usize,
    alpha: DVector<f64>,
}

impl GpuSeparableProblem {
    /// == SeparableProblemBuilder::{new|mrhs, observations, weights, epsilon, build}
    pub fn build(desc: vp_model_desc, x: &DVector<f64>, y: &DMatrix<f64>, w: Option<&DVector<f64>>, eps: Option<f64>,
                 initial: DVector<f64>) -> Result<Self, String> {
        let (m, s) = (y.nrows(), y.ncols());
        let mut h: *mut vp_batch = null_mut();
        let rc = unsafe {
            vp_batch_create(&mut h, &desc, VP_F64, m as i64, s as i64, 1, x.as_ptr() as *const c_void,
                y.as_ptr() as *const c_void, w.map_or(std::ptr::null(), |w| w.as_ptr() as *const c_void),
                eps.unwrap_or(-1.0), VP_FLAG_OWN_STREAM, 0, null_mut())
        };
        if rc != 0 { return Err(last_error()); }
        let mut p = Self { h, m, s, n: desc.n_basis as usize, q: desc.n_params as usize, alpha: initial.clone() };
        p.set_params(&initial); // build() runs the first evaluation (src/problem/builder.rs:321)
        Ok(p)
    }
    /// == SeparableProblem::linear_coefficients (n x S, column-major)
    pub fn linear_coefficients(&self) -> Option<DMatrix<f64>> {
        let mut c = DMatrix::<f64>::zeros(self.n, self.s);
        let mut st = 0i32;
        let rc = unsafe { vp_linear_coeffs(self.h, c.as_mut_ptr() as *mut c_void, &mut st) };
        (rc == 0 && st == 0).then_some(c)
    }
}

impl Drop for GpuSeparableProblem {
    fn drop(&mut self) { unsafe { vp_batch_destroy(self.h) } }
}

impl LeastSquaresProblem<f64, Dyn, Dyn> for GpuSeparableProblem {
    type ResidualStorage = Owned<f64, Dyn>;
    type JacobianStorage = Owned<f64, Dyn, Dyn>;
    type ParameterStorage = Owned<f64, Dyn>;

    fn set_params(&mut self, p: &DVector<f64>) {
        self.alpha = p.clone();
        // failures are latched per problem in the handle, exactly like `cached = None`
        unsafe { vp_set_params(self.h, p.as_ptr() as *const c_void); }
    }
    fn params(&self) -> DVector<f64> { self.alpha.clone() }
    fn residuals(&self) -> Option<DVector<f64>> {
        let mut r = DVector::<f64>::zeros(self.m * self.s);
        let mut st = 0i32;
        let rc = unsafe { vp_residuals(self.h, r.as_mut_ptr() as *mut c_void, &mut st) };
        (rc == 0 && st == 0).then_some(r)
    }
    fn jacobian(&self) -> Option<DMatrix<f64>> {
        // [q][S][m] in memory == (m*S) x q column-major (src/solvers/levmar/mod.rs:147-153)
        let mut j = DMatrix::<f64>::zeros(self.m * self.s, self.q);
        let mut st = 0i32;
        let rc = unsafe { vp_jacobian(self.h, j.as_mut_ptr() as *mut c_void, &mut st) };
        (rc == 0 && st == 0).then_some(j)
    }
}

/// The model plugin over the descriptor language (replaces closures).
pub struct GpuModel { pub desc: vp_model_desc, pub x: DVector<f64>, pub alpha: DVector<f64>, h: *mut vp_batch }

#[derive(Debug, thiserror::Error)]
#[error("varpro_hip: {0}")]
pub struct GpuModelError(String);

impl SeparableNonlinearModel for GpuModel {
    type ScalarType = f64;
    type Error = GpuModelError;
    fn parameter_count(&self) -> usize { self.desc.n_params as usize }
    fn base_function_count(&self) -> usize { self.desc.n_basis as usize }
    fn output_len(&self) -> usize { self.x.len() }
    fn set_params(&mut self, p: DVector<f64>) -> Result<(), Self::Error> { self.alpha = p; Ok(()) }
    fn params(&self) -> DVector<f64> { self.alpha.clone() }
    fn eval(&self) -> Result<DMatrix<f64>, Self::Error> {
        let mut phi = DMatrix::<f64>::zeros(self.x.len(), self.desc.n_basis as usize);
        let rc = unsafe { vp_basis(self.h, self.alpha.as_ptr() as *const c_void, phi.as_mut_ptr() as *mut c_void, null_mut(), 0) };
        if rc != 0 { return Err(GpuModelError(last_error())); }
        Ok(phi)
    }
    fn eval_partial_deriv(&self, k: usize) -> Result<DMatrix<f64>, Self::Error> {
        // dPhi comes back as one column per (basis, parameter) dependency pair in model order; scatter the pairs
        // of parameter k into a zero m x n matrix (zero columns for independent basis functions)
        let (m, n) = (self.x.len(), self.desc.n_basis as usize);
        let pairs: Vec<(usize, usize)> = (0..n).flat_map(|j| (0..2).filter_map(move |a| Some((j, a))))
            .filter(|&(j, a)| self.desc.param[j][a] >= 0).collect();
        let mut dphi = DMatrix::<f64>::zeros(m, pairs.len());
        let rc = unsafe { vp_basis(self.h, self.alpha.as_ptr() as *const c_void, null_mut(), dphi.as_mut_ptr() as *mut c_void, 0) };
        if rc != 0 { return Err(GpuModelError(last_error())); }
        let mut out = DMatrix::<f64>::zeros(m, n);
        for (p, &(j, a)) in pairs.iter().enumerate() {
            if self.desc.param[j][a] as usize == k { out.column_mut(j).axpy(1.0, &dphi.column(p), 1.0); }
        }
        Ok(out)
    }
}

/// Device-resident batched fit: B independent problems, alpha [B][q] in/out, returns one report per problem.
/// `termination > 0`  <=>  `TerminationReason::was_successful()` (src/fit.rs:120-122).
pub fn fit_batch(h: *mut vp_batch, opts: &vp_lm_opts, alpha: &mut [f64], c_out: &mut [f64], b: usize) -> Vec<vp_report> {
    let mut rep = vec![vp_report::default(); b];
    unsafe { vp_fit(h, opts, alpha.as_mut_ptr() as *mut c_void, c_out.as_mut_ptr() as *mut c_void, rep.as_mut_ptr()); }
    rep
}

// ---- round 4: models outside the descriptor language (caller-evaluated Phi / dPhi) and the cost reduction ---------------
extern "C" {
    pub fn vp_batch_create_external(h: *mut *mut vp_batch, n_basis: i32, n_params: i32, n_pairs: i32, pair_basis: *const i32,
                                    pair_param: *const i32, dtype: i32, m: i64, s: i64, b: i64, y: *const c_void,
                                    w: *const c_void, svd_epsilon: f64, flags: i32, device: i32, hip_stream: *mut c_void) -> i32;
    pub fn vp_set_params_with_basis(h: *mut vp_batch, alpha: *const c_void, phi: *const c_void, dphi: *const c_void) -> i32;
    pub fn vp_jacobian_with_derivatives(h: *mut vp_batch, dphi: *const c_void, j_out: *mut c_void, status: *mut i32) -> i32;
    pub fn vp_evaluate_with_basis(h: *mut vp_batch, alpha: *const c_void, phi: *const c_void, dphi: *const c_void,
                                  r: *mut c_void, j: *mut c_void, c: *mut c_void, cost: *mut f64, status: *mut i32) -> i32;
    pub fn vp_reduce_cost(h: *mut vp_batch, rccl_comm: *mut c_void, out4: *mut f64) -> i32;
    // round 5: LevMarSolver::fit for a BATCH of caller-evaluated models by reverse communication
    pub fn vp_fit_begin(h: *mut vp_batch, opts: *const vp_lm_opts, alpha0: *const c_void, flags: i32) -> i32;
    pub fn vp_fit_step_with_basis(h: *mut vp_batch, phi: *const c_void, dphi: *const c_void, alpha_trial_out: *mut c_void,
                                  want_out: *mut i32, n_active_out: *mut i64) -> i32;
    pub fn vp_fit_end(h: *mut vp_batch, alpha_out: *mut c_void, c_out: *mut c_void, rep: *mut vp_report) -> i32;
    pub fn vp_fit_active_set(h: *mut vp_batch, index_out: *mut i32, count_out: *mut i32) -> i32;
}
pub const VP_FIT_DERIVATIVES_ON_ACCEPT: i32 = 1;
pub const VP_WANT_BASIS: i32 = 1;
pub const VP_WANT_DERIVATIVES: i32 = 2;

/// `SeparableProblem<M, Rhs>` for ANY model `M` (src/problem.rs:57-83): the model is evaluated where it lives (host
/// closures, `src/model/mod.rs:441-512`), everything downstream of `eval()` / `eval_partial_deriv(k)` runs on the device
/// (weighting src/solvers/levmar/mod.rs:47,141; solve + residual :51-59; Kaufman Jacobian :101-201).
pub struct GpuProblemAnyModel<M: SeparableNonlinearModel<ScalarType = f64>> {
    model: M,
    h: *mut vp_batch,
    pairs: Vec<(usize, usize)>, // (basis j, parameter k) whose derivative is not identically zero, fixed at build()
    m: usize,
    s: usize,
}

impl<M: SeparableNonlinearModel<ScalarType = f64>> GpuProblemAnyModel<M> {
    /// == SeparableProblemBuilder::build (src/problem/builder.rs:278-324)
    pub fn build(model: M, y: &DMatrix<f64>, w: Option<&DVector<f64>>, eps: f64, pairs: Vec<(usize, usize)>) -> Result<Self, String> {
        let (pb, pp): (Vec<i32>, Vec<i32>) = pairs.iter().map(|&(j, k)| (j as i32, k as i32)).unzip();
        let mut h: *mut vp_batch = null_mut();
        let rc = unsafe {
            vp_batch_create_external(&mut h, model.base_function_count() as i32, model.parameter_count() as i32, pairs.len() as i32,
                                     pb.as_ptr(), pp.as_ptr(), 0 /* VP_F64 */, y.nrows() as i64, y.ncols() as i64, 1,
                                     y.as_ptr() as *const c_void, w.map_or(std::ptr::null(), |w| w.as_ptr() as *const c_void), eps,
                                     0, 0, null_mut())
        };
        if rc != 0 { return Err(last_error()); }
        let mut p = Self { model, h, pairs, m: y.nrows(), s: y.ncols() };
        let a0 = p.model.params();
        p.set_params(&a0); // the builder's initial evaluation (src/problem/builder.rs:321)
        Ok(p)
    }
}

impl<M: SeparableNonlinearModel<ScalarType = f64>> Drop for GpuProblemAnyModel<M> {
    fn drop(&mut self) { unsafe { vp_batch_destroy(self.h) } }
}

impl<M: SeparableNonlinearModel<ScalarType = f64>> LeastSquaresProblem<f64, Dyn, Dyn> for GpuProblemAnyModel<M> {
    type ResidualStorage = Owned<f64, Dyn>;
    type JacobianStorage = Owned<f64, Dyn, Dyn>;
    type ParameterStorage = Owned<f64, Dyn>;

    fn set_params(&mut self, x: &DVector<f64>) { // src/solvers/levmar/mod.rs:42-73
        let phi = if self.model.set_params(x.clone()).is_ok() { self.model.eval().ok() } else { None };
        // a model error is handed over as a NaN basis: status != 0  <=>  cached = None (:61-72)
        let phi = phi.unwrap_or_else(|| DMatrix::from_element(self.m, self.model.base_function_count(), f64::NAN));
        unsafe { vp_set_params_with_basis(self.h, x.as_ptr() as *const c_void, phi.as_ptr() as *const c_void, std::ptr::null()); }
    }
    fn params(&self) -> DVector<f64> { self.model.params() }
    fn residuals(&self) -> Option<DVector<f64>> { // :91-95
        let (mut r, mut st) = (DVector::<f64>::zeros(self.m * self.s), 0i32);
        let rc = unsafe { vp_residuals(self.h, r.as_mut_ptr() as *mut c_void, &mut st) };
        (rc == 0 && st == 0).then_some(r)
    }
    fn jacobian(&self) -> Option<DMatrix<f64>> { // :101-201; eval_partial_deriv(k) is only needed HERE (:141)
        let q = self.model.parameter_count();
        let mut dphi = vec![0f64; self.pairs.len() * self.m]; // the non-zero columns, in pair order
        let mut cache: Vec<Option<DMatrix<f64>>> = vec![None; q];
        for (p, &(j, k)) in self.pairs.iter().enumerate() {
            if cache[k].is_none() { cache[k] = Some(self.model.eval_partial_deriv(k).ok()?); }
            dphi[p * self.m..(p + 1) * self.m].copy_from_slice(cache[k].as_ref().unwrap().column(j).as_slice());
        }
        let (mut jac, mut st) = (DMatrix::<f64>::zeros(self.m * self.s, q), 0i32);
        let rc = unsafe { vp_jacobian_with_derivatives(self.h, dphi.as_ptr() as *const c_void, jac.as_mut_ptr() as *mut c_void, &mut st) };
        (rc == 0 && st == 0).then_some(jac)
    }
}
