//! Known-answer vectors for border_amd's oracle from the REAL crates border pins - the three things a machine without
//! cargo cannot produce (VERDICT r2 "missing" #3):
//!
//!  1. `StdRng::seed_from_u64(seed)` output words (rand =0.8.5 -> rand_chacha ChaCha12 + rand_core's PCG32 seed expansion) and
//!     the index stream of `SimpleReplayBuffer::batch` built from them
//!     (`border-core/src/generic_replay_buffer/base.rs:353, 384-390`: `(rng.next_u32() as usize) % size`);
//!  2. a file written by `tch::nn::VarStore::save` (tch 0.16 / libtorch 2.3.0), the container of `qnet.pt.tch`
//!     (`border-tch-agent/src/dqn/model/base.rs:134-148`), and its safetensors twin;
//!  3. `image::imageops::resize(.., 84, 84, Triangle)` + the luma expression of `BorderAtariEnv::warp_and_grayscale`
//!     (`border-atari-env/src/env.rs:171-195`) on a 210x160 frame both sides generate from the same integer formula.
//!
//! Usage:  cargo run --release [-- <out dir>]      (default ./out)
//! Then:   python tests/golden/ingest_upstream.py <out dir> [--accept]     in the border_amd repository.
use image::{imageops::resize, imageops::FilterType::Triangle, ImageBuffer, Rgb};
use rand::{rngs::StdRng, RngCore, SeedableRng};
use std::{fmt::Write as _, fs, path::PathBuf};
use tch::{nn, Device, Kind, Tensor};

const SEEDS: [u64; 6] = [42, 0, 1, 43, 49, 0x8000_0000_0000_0005];
const N_U32: usize = 64;
const N_U64: usize = 4;
const INDEX_SIZE: usize = 1_000_000;
const INDEX_N: usize = 768; // three batches of 256

/// The 210x160 RGB test frame: flat "playfield" bands, a few "sprites", two noisy rows - every value from integer arithmetic
/// that `tests/golden/ingest_upstream.py::kat_frame` repeats exactly.
fn kat_frame() -> Vec<u8> {
    let (w, h) = (160usize, 210usize);
    let palette: [[u8; 3]; 7] = [[0, 0, 0], [200, 72, 72], [45, 50, 184], [72, 160, 72], [214, 214, 214], [252, 188, 116], [84, 138, 210]];
    let mut f = vec![0u8; w * h * 3];
    let mut lcg: u32 = 12345;
    for y in 0..h {
        for x in 0..w {
            let mut c = palette[(y / 15 + x / 20) % 7];
            // sprites: 8x8 blocks on a lattice
            if (y % 37) < 8 && (x % 29) < 8 {
                c = palette[(y / 37 + x / 29 + 3) % 7];
            }
            // one-pixel lines (what a Triangle filter smears)
            if y == 100 || x == 77 {
                c = [255, 255, 255];
            }
            for k in 0..3 {
                let mut v = c[k];
                if y == 50 || y == 151 {
                    lcg = lcg.wrapping_mul(1664525).wrapping_add(1013904223);
                    v = (lcg >> 24) as u8;
                }
                f[(y * w + x) * 3 + k] = v;
            }
        }
    }
    f
}

fn json_list<T: std::fmt::Display>(v: &[T]) -> String {
    let mut s = String::from("[");
    for (i, x) in v.iter().enumerate() {
        if i > 0 {
            s.push_str(", ");
        }
        write!(s, "{}", x).unwrap();
    }
    s.push(']');
    s
}

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let out: PathBuf = std::env::args().nth(1).unwrap_or_else(|| "out".to_string()).into();
    fs::create_dir_all(&out)?;
    let mut j = String::from("{\n");
    writeln!(j, "  \"format\": 1,")?;
    writeln!(j, "  \"crates\": {{\"rand\": \"0.8.5\", \"tch\": \"0.16\", \"image\": \"0.23.14\"}},")?;

    // ---- 1. StdRng ---------------------------------------------------------------------------------------------------------
    writeln!(j, "  \"rng\": [")?;
    for (k, &seed) in SEEDS.iter().enumerate() {
        let mut r = StdRng::seed_from_u64(seed);
        let u32s: Vec<u32> = (0..N_U32).map(|_| r.next_u32()).collect();
        let u64s: Vec<u64> = (0..N_U64).map(|_| r.next_u64()).collect();
        let mut bytes = [0u8; 13]; // fill_bytes with a length that is not a multiple of 4
        r.fill_bytes(&mut bytes);
        let after: u32 = r.next_u32();
        // the replay buffer's draw, from a fresh generator (base.rs:384-390)
        let mut r2 = StdRng::seed_from_u64(seed);
        let ixs: Vec<usize> = (0..INDEX_N).map(|_| (r2.next_u32() as usize) % INDEX_SIZE).collect();
        writeln!(
            j,
            "    {{\"seed\": {}, \"next_u32\": {}, \"then_next_u64\": {}, \"then_fill_bytes_13\": {}, \"then_next_u32\": {}, \"index_size\": {}, \"indices\": {}}}{}",
            seed,
            json_list(&u32s),
            json_list(&u64s),
            json_list(&bytes.to_vec()),
            after,
            INDEX_SIZE,
            json_list(&ixs),
            if k + 1 < SEEDS.len() { "," } else { "" }
        )?;
    }
    writeln!(j, "  ],")?;

    // ---- 2. VarStore files ---------------------------------------------------------------------------------------------------
    // three variables with the reference's naming scheme (sub-path / name -> "c1.weight"), values = index * 0.25 - 3 (exact in f32)
    {
        let vs = nn::VarStore::new(Device::Cpu);
        let root = vs.root();
        let mk = |n: i64, off: i64| Tensor::arange(n, (Kind::Float, Device::Cpu)) * 0.25 - 3.0 + off as f64;
        let _w = (&root / "c1").var_copy("weight", &mk(2 * 3 * 2 * 2, 0).reshape([2, 3, 2, 2]));
        let _b = (&root / "c1").var_copy("bias", &mk(2, 100));
        let _l = (&root / "l2").var_copy("weight", &mk(3 * 4, 200).reshape([3, 4]));
        vs.save(out.join("varstore.pt.tch"))?; // what DqnModel::save does for "qnet.pt.tch"
        vs.save(out.join("varstore.safetensors"))?; // VarStore's other container (chosen by the extension)
    }
    writeln!(
        j,
        "  \"varstore\": {{\"files\": [\"varstore.pt.tch\", \"varstore.safetensors\"], \"tensors\": [[\"c1.weight\", [2, 3, 2, 2], 0], [\"c1.bias\", [2], 100], [\"l2.weight\", [3, 4], 200]], \"formula\": \"value[i] = i * 0.25 - 3 + offset\"}},"
    )?;

    // ---- 3. image 0.23.14 Triangle resize + border's luma ------------------------------------------------------------------------
    {
        let frame = kat_frame();
        let img = ImageBuffer::<Rgb<u8>, Vec<u8>>::from_vec(160, 210, frame).expect("frame size");
        let small = resize(&img, 84, 84, Triangle);
        let rgb = small.to_vec();
        // env.rs:176-186 names the channels (b, g, r) in buffer order and weighs them .114 / .587 / .299
        let gray: Vec<u8> = rgb.chunks_exact(3).map(|p| ((0.299 * p[2] as f32) + (0.587 * p[1] as f32) + (0.114 * p[0] as f32)) as u8).collect();
        writeln!(j, "  \"resize\": {{\"width\": 160, \"height\": 210, \"rgb_84x84\": {}, \"gray_84x84\": {}}}", json_list(&rgb), json_list(&gray))?;
    }
    j.push_str("}\n");
    fs::write(out.join("upstream_kat.json"), j)?;
    println!("wrote {}", out.join("upstream_kat.json").display());
    Ok(())
}
