//! Agent configurations with the reference's field names, defaults, builder methods and serde layout, minus the tch type
//! parameters: `border-tch-agent/src/{dqn/config.rs:26-48, dqn/model/config.rs, iqn/config.rs, iqn/model/config.rs,
//! sac/config.rs, sac/actor/config.rs, sac/critic/config.rs, sac/ent_coef.rs:10-15, opt.rs:13-28, util.rs:17-23,
//! mlp/config.rs:7-12, cnn/config.rs:13-18, dqn/explorer.rs:8-46, iqn/explorer.rs:9-44, lib.rs:19-25}`.
//! A YAML file written by border-tch-agent's `DqnConfig::save` loads here (`PhantomData` fields serialise as `null` and are
//! ignored), and `fill()` copies the fields into the C structs of `include/border_amd.h`.
use crate::ffi;
use anyhow::{anyhow, Result};
use serde::{Deserialize, Serialize};
use std::{
    fs::File,
    io::{BufReader, Write},
    path::Path,
};

/// `border_tch_agent::Device`; `Cuda(n)` names HIP device `n` (the reference's own spelling is kept so configs load).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone, Copy)]
pub enum Device {
    Cpu,
    Cuda(usize),
}

impl Device {
    /// HIP ordinal; `Cpu` has no implementation here (north_star: the path runs on the GPU, there is no CPU fallback).
    pub fn ordinal(dev: &Option<Device>, agent: &str) -> i32 {
        match dev {
            None => panic!("No device is given for {} agent", agent), // dqn/base.rs:256-259
            Some(Device::Cpu) => panic!("border-amd-agent has no CPU path: use Device::Cuda(n) (HIP device n)"),
            Some(Device::Cuda(n)) => *n as i32,
        }
    }
}

/// `util::CriticLoss`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone, Copy)]
pub enum CriticLoss {
    Mse,
    SmoothL1,
}

impl CriticLoss {
    pub fn code(&self) -> i32 {
        match self {
            CriticLoss::Mse => ffi::BDR_LOSS_MSE,
            CriticLoss::SmoothL1 => ffi::BDR_LOSS_SMOOTH_L1,
        }
    }
}

/// `opt::OptimizerConfig`.
#[derive(Debug, Clone, Deserialize, Serialize, PartialEq)]
pub enum OptimizerConfig {
    Adam { lr: f64 },
    AdamW { lr: f64, beta1: f64, beta2: f64, wd: f64, eps: f64, amsgrad: bool },
}

impl OptimizerConfig {
    pub fn lr(&self) -> f64 {
        match self {
            OptimizerConfig::Adam { lr } => *lr,
            OptimizerConfig::AdamW { lr, .. } => *lr,
        }
    }

    /// -> the AdamW fields a `bdr_{iqn,sac}_config` carries beside the model's `lr` (`OptimizerConfig::build`, opt.rs:30-57).
    pub(crate) fn fill(&self, c: &mut ffi::bdr_adamw_config) {
        match self {
            OptimizerConfig::Adam { .. } => c.opt_kind = ffi::BDR_OPT_ADAM, // the library's defaults are tch's Adam::default()
            OptimizerConfig::AdamW { lr: _, beta1, beta2, wd, eps, amsgrad } => {
                c.opt_kind = ffi::BDR_OPT_ADAMW;
                c.beta1 = *beta1;
                c.beta2 = *beta2;
                c.weight_decay = *wd;
                c.eps = *eps;
                c.amsgrad = *amsgrad as i32;
            }
        }
    }
}

/// Which products the large matrix layers compute (`BDR_ARITH_*`, include/border_amd.h).  NOT a reference field - the reference
/// has one arithmetic, f32 - and absent in a reference YAML (-> the library default).  Both settings accumulate in f32 and stay
/// inside the 1e-4 parity bar; `F32Exact` is what to choose when Q-values must not depend on the kernel selection at all.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone, Copy)]
pub enum Arithmetic {
    /// conv2 / conv3 forward (DQN), merge / embedding layers (IQN): operands split into three bf16 terms, six of nine products.
    Bf16x3,
    /// every product an f32 x f32 product on the FP32 MFMA.
    F32Exact,
}

impl Default for Arithmetic {
    fn default() -> Self {
        Arithmetic::Bf16x3
    }
}

impl Arithmetic {
    pub fn code(&self) -> i32 {
        match self {
            Arithmetic::Bf16x3 => ffi::BDR_ARITH_BF16X3_6,
            Arithmetic::F32Exact => ffi::BDR_ARITH_F32_EXACT,
        }
    }
}

/// `mlp::MlpConfig`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct MlpConfig {
    pub in_dim: i64,
    pub units: Vec<i64>,
    pub out_dim: i64,
    pub activation_out: bool,
}

impl MlpConfig {
    pub fn new(in_dim: i64, units: Vec<i64>, out_dim: i64, activation_out: bool) -> Self {
        Self { in_dim, units, out_dim, activation_out }
    }
}

fn default_skip_linear() -> bool {
    false
}

/// `cnn::AtariCnnConfig`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct AtariCnnConfig {
    pub n_stack: i64,
    pub out_dim: i64,
    #[serde(default = "default_skip_linear")]
    pub skip_linear: bool,
}

impl AtariCnnConfig {
    pub fn new(n_stack: i64, out_dim: i64) -> Self {
        Self { n_stack, out_dim, skip_linear: false }
    }

    pub fn skip_linear(mut self, skip_linear: bool) -> Self {
        self.skip_linear = skip_linear;
        self
    }
}

/// The `Q::Config` of `DqnConfig<Q>` / `F::Config` of `IqnConfig<F, M>`: the reference is generic over the sub-model, the
/// library implements the two sub-models the reference ships.  Untagged, so `q_config: {n_stack: 4, out_dim: 0}` and
/// `q_config: {in_dim: 4, units: [256, 256], out_dim: 2, activation_out: false}` both load as the reference writes them.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
#[serde(untagged)]
pub enum QNetConfig {
    AtariCnn(AtariCnnConfig),
    Mlp(MlpConfig),
}

impl QNetConfig {
    pub fn get_out_dim(&self) -> i64 {
        match self {
            QNetConfig::AtariCnn(c) => c.out_dim,
            QNetConfig::Mlp(c) => c.out_dim,
        }
    }

    pub fn set_out_dim(&mut self, v: i64) {
        match self {
            QNetConfig::AtariCnn(c) => c.out_dim = v,
            QNetConfig::Mlp(c) => c.out_dim = v,
        }
    }

    pub(crate) fn fill(&self, net: &mut ffi::bdr_net_config) -> Result<()> {
        match self {
            QNetConfig::AtariCnn(c) => {
                net.kind = ffi::BDR_NET_ATARI_CNN;
                net.n_stack = c.n_stack as i32;
                net.out_dim = c.out_dim as i32;
            }
            QNetConfig::Mlp(c) => {
                if c.units.len() > ffi::BDR_MAX_UNITS {
                    return Err(anyhow!("MlpConfig: at most {} hidden layers", ffi::BDR_MAX_UNITS));
                }
                net.kind = ffi::BDR_NET_MLP;
                net.in_dim = c.in_dim as i32;
                net.n_units = c.units.len() as i32;
                for (i, u) in c.units.iter().enumerate() {
                    net.units[i] = *u as i32;
                }
                net.out_dim = c.out_dim as i32;
                net.activation_out = c.activation_out as i32;
            }
        }
        Ok(())
    }
}

impl From<AtariCnnConfig> for QNetConfig {
    fn from(c: AtariCnnConfig) -> Self {
        QNetConfig::AtariCnn(c)
    }
}

impl From<MlpConfig> for QNetConfig {
    fn from(c: MlpConfig) -> Self {
        QNetConfig::Mlp(c)
    }
}

pub(crate) fn fill_units(units: &[i64], n: &mut i32, out: &mut [i32; 8], what: &str) -> Result<()> {
    if units.len() > ffi::BDR_MAX_UNITS {
        return Err(anyhow!("{}: at most {} hidden layers", what, ffi::BDR_MAX_UNITS));
    }
    *n = units.len() as i32;
    for (i, u) in units.iter().enumerate() {
        out[i] = *u as i32;
    }
    Ok(())
}

// ---------------------------------------------------------------------------------------------- explorers
/// `dqn::explorer::Softmax` / `iqn::explorer::Softmax`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone, Default)]
pub struct Softmax {}

impl Softmax {
    pub fn new() -> Self {
        Self {}
    }
}

/// `dqn::explorer::EpsilonGreedy` (`n_opts` counts `action()` calls, as in the reference).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct EpsilonGreedy {
    pub n_opts: usize,
    pub eps_start: f64,
    pub eps_final: f64,
    pub final_step: usize,
}

impl Default for EpsilonGreedy {
    fn default() -> Self {
        Self { n_opts: 0, eps_start: 1.0, eps_final: 0.02, final_step: 100_000 }
    }
}

impl EpsilonGreedy {
    pub fn new() -> Self {
        Self::default()
    }

    pub fn with_final_step(final_step: usize) -> DqnExplorer {
        DqnExplorer::EpsilonGreedy(Self { final_step, ..Self::default() })
    }
}

/// `dqn::explorer::DqnExplorer`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub enum DqnExplorer {
    Softmax(Softmax),
    EpsilonGreedy(EpsilonGreedy),
}

/// `iqn::explorer::IqnExplorer` (same shape).
pub type IqnExplorer = DqnExplorer;

impl DqnExplorer {
    /// -> `bdr_explorer_config`; `seed` is the library's addition (the reference draws from fastrand's unseeded global).
    pub(crate) fn to_c(&self, seed: u64) -> ffi::bdr_explorer_config {
        let mut e = ffi::bdr_explorer_config { kind: 0, eps_start: 0.0, eps_final: 0.0, final_step: 0, n_calls: 0, seed: 0 };
        match self {
            DqnExplorer::Softmax(_) => unsafe { ffi::bdr_explorer_config_default(&mut e, ffi::BDR_EXPLORER_SOFTMAX) },
            DqnExplorer::EpsilonGreedy(g) => {
                unsafe { ffi::bdr_explorer_config_default(&mut e, ffi::BDR_EXPLORER_EPS_GREEDY) };
                e.eps_start = g.eps_start;
                e.eps_final = g.eps_final;
                e.final_step = g.final_step as u64;
                e.n_calls = g.n_opts as u64;
            }
        }
        e.seed = seed;
        e
    }
}

// ---------------------------------------------------------------------------------------------- DQN
/// `dqn::DqnModelConfig<Q::Config>`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct DqnModelConfig {
    pub q_config: Option<QNetConfig>,
    pub opt_config: OptimizerConfig,
}

impl Default for DqnModelConfig {
    fn default() -> Self {
        Self { q_config: None, opt_config: OptimizerConfig::Adam { lr: 0.0 } }
    }
}

impl DqnModelConfig {
    pub fn q_config(mut self, v: impl Into<QNetConfig>) -> Self {
        self.q_config = Some(v.into());
        self
    }

    pub fn out_dim(mut self, v: i64) -> Self {
        if let Some(q) = &mut self.q_config {
            q.set_out_dim(v);
        }
        self
    }

    pub fn opt_config(mut self, v: OptimizerConfig) -> Self {
        self.opt_config = v;
        self
    }
}

/// `dqn::DqnConfig<Q>` (`dqn/config.rs:26-48`, defaults `:82-102`).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct DqnConfig {
    pub model_config: DqnModelConfig,
    pub soft_update_interval: usize,
    pub n_updates_per_opt: usize,
    pub batch_size: usize,
    pub discount_factor: f64,
    pub tau: f64,
    pub train: bool,
    pub explorer: DqnExplorer,
    #[serde(default)]
    pub clip_reward: Option<f64>,
    #[serde(default)]
    pub double_dqn: bool,
    pub clip_td_err: Option<(f64, f64)>,
    pub device: Option<Device>,
    pub critic_loss: CriticLoss,
    pub record_verbose_level: usize,
    /// Seed of the exploration stream and of the library's parameter initialiser (not in the reference, whose fastrand /
    /// torch generators are unseeded; absent in a reference YAML -> 0).
    #[serde(default)]
    pub seed: u64,
    /// `bdr_dqn_config::arithmetic` (not a reference field; absent in a reference YAML -> the library default).
    #[serde(default)]
    pub arithmetic: Arithmetic,
}

impl Default for DqnConfig {
    fn default() -> Self {
        Self {
            model_config: Default::default(),
            soft_update_interval: 1,
            n_updates_per_opt: 1,
            batch_size: 1,
            discount_factor: 0.99,
            tau: 0.005,
            train: false,
            explorer: DqnExplorer::Softmax(Softmax::new()),
            clip_reward: None,
            double_dqn: false,
            clip_td_err: None,
            device: None,
            critic_loss: CriticLoss::Mse,
            record_verbose_level: 0,
            seed: 0,
            arithmetic: Arithmetic::default(),
        }
    }
}

macro_rules! setter {
    ($name:ident, $t:ty) => {
        pub fn $name(mut self, v: $t) -> Self {
            self.$name = v;
            self
        }
    };
}

macro_rules! yaml_io {
    () => {
        /// `Config::load` of the reference (serde_yaml).
        pub fn load(path: impl AsRef<Path>) -> Result<Self> {
            let rdr = BufReader::new(File::open(path)?);
            Ok(serde_yaml::from_reader(rdr)?)
        }

        /// `Config::save` of the reference.
        pub fn save(&self, path: impl AsRef<Path>) -> Result<()> {
            let mut file = File::create(path)?;
            file.write_all(serde_yaml::to_string(&self)?.as_bytes())?;
            Ok(())
        }
    };
}

impl DqnConfig {
    setter!(soft_update_interval, usize);
    setter!(n_updates_per_opt, usize);
    setter!(batch_size, usize);
    setter!(discount_factor, f64);
    setter!(tau, f64);
    setter!(explorer, DqnExplorer);
    setter!(model_config, DqnModelConfig);
    setter!(double_dqn, bool);
    setter!(critic_loss, CriticLoss);
    setter!(record_verbose_level, usize);
    setter!(seed, u64);
    setter!(arithmetic, Arithmetic);
    yaml_io!();

    pub fn out_dim(mut self, out_dim: i64) -> Self {
        self.model_config = self.model_config.out_dim(out_dim);
        self
    }

    pub fn clip_reward(mut self, clip_reward: Option<f64>) -> Self {
        self.clip_reward = clip_reward; // stored, never used: `_clip_reward` in dqn/base.rs:42,276
        self
    }

    pub fn clip_td_err(mut self, clip_td_err: Option<(f64, f64)>) -> Self {
        self.clip_td_err = clip_td_err;
        self
    }

    pub fn device(mut self, device: Device) -> Self {
        self.device = Some(device);
        self
    }

    /// -> `bdr_dqn_config` (same names; `Configurable::build`, dqn/base.rs:252-286).
    pub(crate) fn to_c(&self) -> Result<ffi::bdr_dqn_config> {
        let mut c: ffi::bdr_dqn_config = unsafe { std::mem::zeroed() };
        unsafe { ffi::bdr_dqn_config_default(&mut c) };
        let q = self.model_config.q_config.as_ref().ok_or_else(|| anyhow!("q_config is not set."))?; // dqn/model/base.rs:62
        q.fill(&mut c.net)?;
        match &self.model_config.opt_config {
            OptimizerConfig::Adam { lr } => {
                c.opt_kind = ffi::BDR_OPT_ADAM;
                c.lr = *lr;
            }
            OptimizerConfig::AdamW { lr, beta1, beta2, wd, eps, amsgrad } => {
                c.opt_kind = ffi::BDR_OPT_ADAMW;
                c.lr = *lr;
                c.beta1 = *beta1;
                c.beta2 = *beta2;
                c.weight_decay = *wd;
                c.eps = *eps;
                c.amsgrad = *amsgrad as i32;
            }
        }
        c.soft_update_interval = self.soft_update_interval as u64;
        c.n_updates_per_opt = self.n_updates_per_opt as u64;
        c.batch_size = self.batch_size as u64;
        c.discount_factor = self.discount_factor;
        c.tau = self.tau;
        c.train = self.train as i32;
        c.double_dqn = self.double_dqn as i32;
        c.critic_loss = self.critic_loss.code();
        if let Some((lo, hi)) = self.clip_td_err {
            c.has_clip_td_err = 1;
            c.clip_td_err_min = lo;
            c.clip_td_err_max = hi;
        }
        c.record_verbose_level = self.record_verbose_level as i32;
        c.device = Device::ordinal(&self.device, "DQN");
        c.param_seed = self.seed;
        c.arithmetic = self.arithmetic.code();
        Ok(c)
    }
}

// ---------------------------------------------------------------------------------------------- IQN
/// `iqn::IqnSample` (`iqn/model/base.rs:327-352`).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone, Copy)]
pub enum IqnSample {
    Const10,
    Const32,
    Uniform10,
    Uniform8,
    Uniform32,
    Uniform64,
    Median,
}

impl IqnSample {
    pub fn code(&self) -> i32 {
        match self {
            IqnSample::Const10 => ffi::BDR_IQN_CONST10,
            IqnSample::Const32 => ffi::BDR_IQN_CONST32,
            IqnSample::Uniform10 => ffi::BDR_IQN_UNIFORM10,
            IqnSample::Uniform8 => ffi::BDR_IQN_UNIFORM8,
            IqnSample::Uniform32 => ffi::BDR_IQN_UNIFORM32,
            IqnSample::Uniform64 => ffi::BDR_IQN_UNIFORM64,
            IqnSample::Median => ffi::BDR_IQN_MEDIAN,
        }
    }
}

/// `iqn::IqnModelConfig<F::Config, M::Config>`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct IqnModelConfig {
    pub feature_dim: i64,
    pub embed_dim: i64,
    pub f_config: Option<QNetConfig>,
    pub m_config: Option<MlpConfig>,
    pub opt_config: OptimizerConfig,
}

impl Default for IqnModelConfig {
    fn default() -> Self {
        Self { feature_dim: 0, embed_dim: 0, f_config: None, m_config: None, opt_config: OptimizerConfig::Adam { lr: 0.0 } }
    }
}

impl IqnModelConfig {
    setter!(feature_dim, i64);
    setter!(embed_dim, i64);
    setter!(opt_config, OptimizerConfig);

    pub fn f_config(mut self, v: impl Into<QNetConfig>) -> Self {
        self.f_config = Some(v.into());
        self
    }

    pub fn m_config(mut self, v: MlpConfig) -> Self {
        self.m_config = Some(v);
        self
    }

    pub fn out_dim(mut self, v: i64) -> Self {
        if let Some(m) = &mut self.m_config {
            m.out_dim = v;
        }
        self
    }

    pub fn learning_rate(mut self, v: f64) -> Self {
        match &self.opt_config {
            OptimizerConfig::Adam { lr: _ } => self.opt_config = OptimizerConfig::Adam { lr: v },
            _ => unimplemented!(), // iqn/model/config.rs:18-24
        };
        self
    }
}

/// `iqn::IqnConfig<F, M>` (`iqn/config.rs:18-39`, defaults `:50-67`).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct IqnConfig {
    pub model_config: IqnModelConfig,
    pub soft_update_interval: usize,
    pub n_updates_per_opt: usize,
    pub batch_size: usize,
    pub discount_factor: f64,
    pub tau: f64,
    pub train: bool,
    pub explorer: IqnExplorer,
    pub sample_percents_pred: IqnSample,
    pub sample_percents_tgt: IqnSample,
    pub sample_percents_act: IqnSample,
    pub device: Option<Device>,
    /// Seed of the percent-point stream (`Tensor::rand` in the reference, iqn/model/base.rs:365-368) and of exploration.
    #[serde(default)]
    pub seed: u64,
    /// `bdr_iqn_config::arithmetic` (not a reference field; absent in a reference YAML -> the library default).
    #[serde(default)]
    pub arithmetic: Arithmetic,
}

impl Default for IqnConfig {
    fn default() -> Self {
        Self {
            model_config: Default::default(),
            soft_update_interval: 1,
            n_updates_per_opt: 1,
            batch_size: 1,
            discount_factor: 0.99,
            tau: 0.005,
            sample_percents_pred: IqnSample::Uniform8,
            sample_percents_tgt: IqnSample::Uniform8,
            sample_percents_act: IqnSample::Const32,
            train: false,
            explorer: DqnExplorer::Softmax(Softmax::new()),
            device: None,
            seed: 0,
            arithmetic: Arithmetic::default(),
        }
    }
}

impl IqnConfig {
    setter!(model_config, IqnModelConfig);
    setter!(soft_update_interval, usize);
    setter!(n_updates_per_opt, usize);
    setter!(batch_size, usize);
    setter!(discount_factor, f64);
    setter!(tau, f64);
    setter!(explorer, IqnExplorer);
    setter!(sample_percents_pred, IqnSample);
    setter!(sample_percents_tgt, IqnSample);
    setter!(sample_percents_act, IqnSample);
    setter!(seed, u64);
    setter!(arithmetic, Arithmetic);
    yaml_io!();

    pub fn out_dim(mut self, out_dim: i64) -> Self {
        self.model_config = self.model_config.out_dim(out_dim);
        self
    }

    pub fn learning_rate(mut self, lr: f64) -> Self {
        self.model_config = self.model_config.learning_rate(lr);
        self
    }

    pub fn device(mut self, device: Device) -> Self {
        self.device = Some(device);
        self
    }

    pub(crate) fn to_c(&self) -> Result<ffi::bdr_iqn_config> {
        let mut c: ffi::bdr_iqn_config = unsafe { std::mem::zeroed() };
        unsafe { ffi::bdr_iqn_config_default(&mut c) };
        let m = &self.model_config;
        let f = m.f_config.as_ref().ok_or_else(|| anyhow!("f_config is not set."))?;
        let mm = m.m_config.as_ref().ok_or_else(|| anyhow!("m_config is not set."))?;
        f.fill(&mut c.psi)?;
        if let QNetConfig::AtariCnn(a) = f {
            if !a.skip_linear {
                return Err(anyhow!("IQN's feature extractor is AtariCnn{{skip_linear: true}} (3136 features) or an Mlp"));
            }
        }
        c.psi.out_dim = m.feature_dim as i32;
        c.feature_dim = m.feature_dim as i32;
        c.embed_dim = m.embed_dim as i32;
        if mm.in_dim != m.feature_dim {
            return Err(anyhow!("m_config.in_dim ({}) must equal feature_dim ({})", mm.in_dim, m.feature_dim));
        }
        fill_units(&mm.units, &mut c.n_f_units, &mut c.f_units, "m_config")?;
        c.n_actions = mm.out_dim as i32;
        c.lr = m.opt_config.lr(); // iqn/model/config.rs:50 -> opt.rs:30-57 (Adam or AdamW)
        m.opt_config.fill(&mut c.opt);
        c.arithmetic = self.arithmetic.code();
        c.soft_update_interval = self.soft_update_interval as u64;
        c.n_updates_per_opt = self.n_updates_per_opt as u64;
        c.batch_size = self.batch_size as u64;
        c.discount_factor = self.discount_factor;
        c.tau = self.tau;
        c.sample_percents_pred = self.sample_percents_pred.code();
        c.sample_percents_tgt = self.sample_percents_tgt.code();
        c.sample_percents_act = self.sample_percents_act.code();
        c.train = self.train as i32;
        c.device = Device::ordinal(&self.device, "IQN");
        c.seed = self.seed;
        Ok(c)
    }
}

// ---------------------------------------------------------------------------------------------- SAC
/// `sac::EntCoefMode` (`sac/ent_coef.rs:10-15`): `Fix(alpha)` or `Auto(target_entropy, learning_rate)`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub enum EntCoefMode {
    Fix(f64),
    Auto(f64, f64),
}

/// `sac::ActorConfig<P::Config>` with `P = Mlp2` (`mlp/mlp2.rs`): `pi_config.out_dim` is the action dimension.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct ActorConfig {
    pub pi_config: Option<MlpConfig>,
    pub opt_config: OptimizerConfig,
}

impl Default for ActorConfig {
    fn default() -> Self {
        Self { pi_config: None, opt_config: OptimizerConfig::Adam { lr: 0.0 } }
    }
}

impl ActorConfig {
    setter!(opt_config, OptimizerConfig);

    pub fn pi_config(mut self, v: MlpConfig) -> Self {
        self.pi_config = Some(v);
        self
    }

    pub fn out_dim(mut self, v: i64) -> Self {
        if let Some(p) = &mut self.pi_config {
            p.out_dim = v;
        }
        self
    }
}

/// `sac::CriticConfig<Q::Config>` with `Q = Mlp` on `cat(obs, act)` (`mlp/base.rs:83-107`): `q_config.in_dim = obs_dim + act_dim`.
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct CriticConfig {
    pub q_config: Option<MlpConfig>,
    pub opt_config: OptimizerConfig,
}

impl Default for CriticConfig {
    fn default() -> Self {
        Self { q_config: None, opt_config: OptimizerConfig::Adam { lr: 0.0 } }
    }
}

impl CriticConfig {
    setter!(opt_config, OptimizerConfig);

    pub fn q_config(mut self, v: MlpConfig) -> Self {
        self.q_config = Some(v);
        self
    }
}

/// `sac::SacConfig<Q, P>` (`sac/config.rs:21-45`, defaults `:85-105`).
#[derive(Debug, Deserialize, Serialize, PartialEq, Clone)]
pub struct SacConfig {
    pub actor_config: ActorConfig,
    pub critic_config: CriticConfig,
    pub gamma: f64,
    pub tau: f64,
    pub ent_coef_mode: EntCoefMode,
    pub epsilon: f64,
    pub min_lstd: f64,
    pub max_lstd: f64,
    pub n_updates_per_opt: usize,
    pub batch_size: usize,
    pub train: bool,
    pub critic_loss: CriticLoss,
    pub reward_scale: f32,
    pub n_critics: usize,
    pub seed: Option<i64>,
    pub device: Option<Device>,
}

impl Default for SacConfig {
    fn default() -> Self {
        Self {
            actor_config: Default::default(),
            critic_config: Default::default(),
            gamma: 0.99,
            tau: 0.005,
            ent_coef_mode: EntCoefMode::Fix(1.0),
            epsilon: 1e-4,
            min_lstd: -20.0,
            max_lstd: 2.0,
            n_updates_per_opt: 1,
            batch_size: 1,
            train: false,
            critic_loss: CriticLoss::Mse,
            reward_scale: 1.0,
            n_critics: 1,
            seed: None,
            device: None,
        }
    }
}

impl SacConfig {
    setter!(actor_config, ActorConfig);
    setter!(critic_config, CriticConfig);
    setter!(ent_coef_mode, EntCoefMode);
    setter!(n_updates_per_opt, usize);
    setter!(batch_size, usize);
    setter!(critic_loss, CriticLoss);
    setter!(reward_scale, f32);
    setter!(n_critics, usize);
    setter!(tau, f64);
    yaml_io!();

    pub fn discount_factor(mut self, v: f64) -> Self {
        self.gamma = v; // sac/config.rs: the setter is named discount_factor, the field gamma
        self
    }

    pub fn seed(mut self, v: i64) -> Self {
        self.seed = Some(v);
        self
    }

    pub fn device(mut self, device: Device) -> Self {
        self.device = Some(device);
        self
    }

    pub(crate) fn to_c(&self) -> Result<ffi::bdr_sac_config> {
        let mut c: ffi::bdr_sac_config = unsafe { std::mem::zeroed() };
        unsafe { ffi::bdr_sac_config_default(&mut c) };
        let pi = self.actor_config.pi_config.as_ref().ok_or_else(|| anyhow!("pi_config is not set."))?;
        let q = self.critic_config.q_config.as_ref().ok_or_else(|| anyhow!("q_config is not set."))?;
        c.obs_dim = pi.in_dim as i32;
        c.act_dim = pi.out_dim as i32;
        if q.in_dim != pi.in_dim + pi.out_dim || q.out_dim != 1 {
            return Err(anyhow!("critic q_config must map obs_dim + act_dim = {} inputs to 1 output", pi.in_dim + pi.out_dim));
        }
        fill_units(&pi.units, &mut c.n_pi_units, &mut c.pi_units, "pi_config")?;
        fill_units(&q.units, &mut c.n_q_units, &mut c.q_units, "q_config")?;
        // sac/actor/config.rs:15, sac/critic/config.rs: each model's own OptimizerConfig (Adam or AdamW, opt.rs:30-57)
        c.lr_actor = self.actor_config.opt_config.lr();
        self.actor_config.opt_config.fill(&mut c.opt_actor);
        c.lr_critic = self.critic_config.opt_config.lr();
        self.critic_config.opt_config.fill(&mut c.opt_critic);
        c.gamma = self.gamma;
        c.tau = self.tau;
        match &self.ent_coef_mode {
            EntCoefMode::Fix(alpha) => {
                c.ent_coef_auto = 0;
                c.ent_coef_alpha = *alpha;
            }
            EntCoefMode::Auto(target_entropy, lr) => {
                c.ent_coef_auto = 1;
                c.target_entropy = *target_entropy;
                c.ent_coef_lr = *lr;
            }
        }
        c.epsilon = self.epsilon;
        c.min_lstd = self.min_lstd;
        c.max_lstd = self.max_lstd;
        c.n_updates_per_opt = self.n_updates_per_opt as u64;
        c.batch_size = self.batch_size as u64;
        c.train = self.train as i32;
        c.critic_loss = self.critic_loss.code();
        c.reward_scale = self.reward_scale as f64;
        c.n_critics = self.n_critics as i32;
        c.device = Device::ordinal(&self.device, "SAC");
        c.seed = self.seed.unwrap_or(0) as u64;
        Ok(c)
    }
}
