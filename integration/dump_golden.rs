// dump_golden.rs -- the recipe that turns "parity unpinned" into a reference-held pin, for a maintainer of pagmerek/frave who has cargo.
// UNVERIFIED SOURCE: written where no Rust toolchain exists (like stages.rs); it uses nothing but std and the crate's own items.
//
// What it does: a `#[test]` that runs the REFERENCE's own code - WaveletImage::from_raster (stages/wavelet_transform.rs:405-432),
// quantization::encode (stages/quantization.rs:7-25), get_lf_context_bucket / get_hf_context_bucket (stages/prediction.rs:86-207) with the dyadic
// parameters of SURVEY.md section 8c, pack_signed (utils.rs:34-40), then the real prediction::encode with its own fit (stages/prediction.rs:224-323)
// and RasterImage::from_wavelet (stages/wavelet_transform.rs:308-356) - on the three known-answer images of SURVEY.md section 8c
// (pixel(x, y, c) = (7x + 13y + 29c + (x*y mod 11)) & 0xFF at 10x10, 64x48 and 100x37, RGB) and writes every array the parity tests compare into
// `ref_kat_<w>x<h>_rgb.npz`, in the layout of this repository's tests/golden/*.npz (tests/golden/make_golden.py). Nothing of the build under test
// is involved: the files are outputs of libfri alone.
//
// How to use it (in a checkout of pagmerek/frave):
//   1. cp dump_golden.rs crates/libfri/src/dump_golden.rs
//   2. add to crates/libfri/src/lib.rs:      #[cfg(test)] mod dump_golden;
//   3. FRI_GOLDEN_DIR=/path/to/frave_amd/tests/golden cargo test -p libfri --release dump_golden -- --nocapture
//      (release: debug builds of libfri panic on integer overflow in entropy_coding.rs, which prediction::encode reaches through finalize_context)
//   4. in frave_amd:  python -m pytest tests/test_golden.py        (tests/golden/ref_*.npz are picked up: the CPU oracle and, with -m gpu, the HIP
//      path must reproduce every array bit for bit; the fitted parameters are compared as GIVEN parameters - the fit itself is an SVD in the
//      reference and normal equations here, its parameters are transmitted and need not match).
// prediction::encode writes ./mse/errors_<channel>.mse into the working directory (emit_mse, stages/prediction.rs:30-37): harmless.

use std::collections::HashMap;
use std::io::Write;

use num::Complex;

use crate::encoder::EncoderOpts;
use crate::images::{ColorSpace, FractalVariant, ImageMetadata, RasterImage};
use crate::stages::prediction::{get_hf_context_bucket, get_lf_context_bucket, CONTEXT_AMOUNT};
use crate::stages::wavelet_transform::{Fractal, WaveletImage};
use crate::stages::{prediction, quantization};
use crate::utils;

const NONE: i32 = i32::MIN; // wire encoding of Option::None (include/fri_hip.h: FRI_HIP_NONE)
const NODES: usize = 512;
const ALPHABET: usize = 1024;

fn kat_image(w: u32, h: u32) -> Vec<u8> {
    let mut v = Vec::with_capacity((w * h * 3) as usize);
    for y in 0..h as u64 {
        for x in 0..w as u64 {
            for c in 0..3u64 {
                v.push(((7 * x + 13 * y + 29 * c + ((x * y) % 11)) & 0xFF) as u8);
            }
        }
    }
    v
}

fn raster(w: u32, h: u32) -> RasterImage {
    RasterImage {
        data: kat_image(w, h),
        metadata: ImageMetadata { height: h, width: w, colorspace: ColorSpace::RGB, variant: FractalVariant::TameTwindragon },
    }
}

/// retained cell centres in canonical order: ascending im, then re (utils.rs:17-32)
fn canonical_centers(lattice: &HashMap<Complex<i32>, Fractal>) -> Vec<Complex<i32>> {
    let mut c: Vec<Complex<i32>> = lattice.keys().cloned().collect();
    c.sort_by(|a, b| utils::order_complex(a, b));
    c
}

fn coefficient_array(image: &WaveletImage, centers: &[Complex<i32>]) -> Vec<i32> {
    let f = centers.len();
    let mut out = vec![NONE; 3 * f * NODES];
    for ch in 0..3 {
        for (k, c) in centers.iter().enumerate() {
            let frac = &image.fractal_lattice[c];
            for i in 0..NODES {
                if let Some(v) = frac.coefficients[ch][i] {
                    out[(ch * f + k) * NODES + i] = v;
                }
            }
        }
    }
    out
}

/// (bucket, prediction) of every Some node of one channel with GIVEN parameters: the calls of prediction::encode's three scans (:237-298), node by node.
/// The context of a node depends on coefficients only, never on predictors computed earlier, so the order of the walk does not matter.
fn predict_given(image: &WaveletImage, centers: &[Complex<i32>], ch: usize, vp: &Vec<[f32; 6]>, wp: &Vec<[f32; 6]>) -> (Vec<u8>, Vec<i32>, Vec<u32>, u64) {
    let f = centers.len();
    let (mut bucket, mut pred) = (vec![0u8; f * NODES], vec![0i32; f * NODES]);
    let (mut hist, mut oob) = (vec![0u32; CONTEXT_AMOUNT * ALPHABET], 0u64);
    for (k, c) in centers.iter().enumerate() {
        let frac = &image.fractal_lattice[c];
        for i in 0..NODES {
            if let Some(value) = frac.coefficients[ch][i] {
                let (b, p) = if i < 2 {
                    get_lf_context_bucket(i, 0, c, &image.fractal_lattice, ch)
                } else {
                    let level = (usize::BITS - 1 - i.leading_zeros()) as u8; // heap index 2^level .. 2^(level+1) - 1
                    get_hf_context_bucket(frac.image_positions[i], level, c, &image.fractal_lattice, &image.global_position_map, vp, wp, ch)
                };
                bucket[k * NODES + i] = b as u8;
                pred[k * NODES + i] = p;
                let sym = utils::pack_signed(value.wrapping_sub(p)) as usize;
                if sym < ALPHABET { hist[b * ALPHABET + sym] += 1 } else { oob += 1 } // bump_freq would index out of bounds (entropy_coding.rs:98-100)
            }
        }
    }
    (bucket, pred, hist, oob)
}

fn predictor_arrays(image: &WaveletImage, centers: &[Complex<i32>], ch: usize) -> (Vec<u8>, Vec<i32>) {
    let f = centers.len();
    let (mut bucket, mut pred) = (vec![0u8; f * NODES], vec![0i32; f * NODES]);
    for (k, c) in centers.iter().enumerate() {
        let frac = &image.fractal_lattice[c];
        for i in 0..NODES {
            let (b, p) = frac.parameter_predictors[ch][i];
            bucket[k * NODES + i] = b as u8;
            pred[k * NODES + i] = p;
        }
    }
    (bucket, pred)
}

// ---- .npy inside an uncompressed .zip = .npz, std only -----------------------------------------------------------------------------------
fn npy(descr: &str, shape: &[usize], raw: &[u8]) -> Vec<u8> {
    let dims = match shape.len() {
        0 => String::from("()"),
        1 => format!("({},)", shape[0]),
        _ => format!("({})", shape.iter().map(|d| d.to_string()).collect::<Vec<_>>().join(", ")),
    };
    let mut header = format!("{{'descr': '{}', 'fortran_order': False, 'shape': {}, }}", descr, dims);
    while (10 + header.len() + 1) % 64 != 0 { header.push(' '); }
    header.push('\n');
    let mut out = Vec::with_capacity(10 + header.len() + raw.len());
    out.extend_from_slice(b"\x93NUMPY\x01\x00");
    out.extend_from_slice(&(header.len() as u16).to_le_bytes());
    out.extend_from_slice(header.as_bytes());
    out.extend_from_slice(raw);
    out
}
fn bytes_i32(v: &[i32]) -> Vec<u8> { v.iter().flat_map(|x| x.to_le_bytes()).collect() }
fn bytes_u32(v: &[u32]) -> Vec<u8> { v.iter().flat_map(|x| x.to_le_bytes()).collect() }
fn bytes_f32(v: &[f32]) -> Vec<u8> { v.iter().flat_map(|x| x.to_le_bytes()).collect() }
fn params_flat(p: &Vec<[f32; 6]>) -> Vec<f32> { p.iter().flat_map(|r| r.iter().cloned()).collect() }

fn crc32(data: &[u8]) -> u32 {
    let mut table = [0u32; 256];
    for i in 0..256u32 {
        let mut c = i;
        for _ in 0..8 { c = if c & 1 != 0 { 0xEDB88320 ^ (c >> 1) } else { c >> 1 }; }
        table[i as usize] = c;
    }
    let mut crc = 0xFFFF_FFFFu32;
    for &b in data { crc = table[((crc ^ b as u32) & 0xFF) as usize] ^ (crc >> 8); }
    crc ^ 0xFFFF_FFFF
}

struct Npz { entries: Vec<(String, Vec<u8>)> }
impl Npz {
    fn new() -> Self { Npz { entries: vec![] } }
    fn add(&mut self, name: &str, descr: &str, shape: &[usize], raw: Vec<u8>) { self.entries.push((format!("{}.npy", name), npy(descr, shape, &raw))); }
    fn scalar_i64(&mut self, name: &str, v: i64) { self.add(name, "<i8", &[], v.to_le_bytes().to_vec()); }
    fn scalar_u64(&mut self, name: &str, v: u64) { self.add(name, "<u8", &[], v.to_le_bytes().to_vec()); }
    fn write(&self, path: &std::path::Path) {
        let mut f = std::fs::File::create(path).expect("cannot create the .npz");
        let (mut central, mut offset) = (Vec::<u8>::new(), 0u32);
        for (name, data) in &self.entries {
            let (crc, size) = (crc32(data), data.len() as u32);
            let mut local = Vec::new();
            local.extend_from_slice(&0x04034b50u32.to_le_bytes());
            for v in [20u16, 0, 0, 0, 0x21] { local.extend_from_slice(&v.to_le_bytes()); } // version, flags, stored, time, date 1980-01-01
            for v in [crc, size, size] { local.extend_from_slice(&v.to_le_bytes()); }
            local.extend_from_slice(&(name.len() as u16).to_le_bytes());
            local.extend_from_slice(&0u16.to_le_bytes());
            local.extend_from_slice(name.as_bytes());
            f.write_all(&local).unwrap();
            f.write_all(data).unwrap();
            central.extend_from_slice(&0x02014b50u32.to_le_bytes());
            for v in [20u16, 20, 0, 0, 0, 0x21] { central.extend_from_slice(&v.to_le_bytes()); }
            for v in [crc, size, size] { central.extend_from_slice(&v.to_le_bytes()); }
            central.extend_from_slice(&(name.len() as u16).to_le_bytes());
            for v in [0u16, 0, 0, 0] { central.extend_from_slice(&v.to_le_bytes()); } // extra, comment, disk, internal attributes
            central.extend_from_slice(&0u32.to_le_bytes()); // external attributes
            central.extend_from_slice(&offset.to_le_bytes());
            central.extend_from_slice(name.as_bytes());
            offset += local.len() as u32 + size;
        }
        f.write_all(&central).unwrap();
        let mut end = Vec::new();
        end.extend_from_slice(&0x06054b50u32.to_le_bytes());
        for v in [0u16, 0, self.entries.len() as u16, self.entries.len() as u16] { end.extend_from_slice(&v.to_le_bytes()); }
        end.extend_from_slice(&(central.len() as u32).to_le_bytes());
        end.extend_from_slice(&offset.to_le_bytes());
        end.extend_from_slice(&0u16.to_le_bytes());
        f.write_all(&end).unwrap();
    }
}

fn dump(w: u32, h: u32, dir: &std::path::Path) {
    // the dyadic parameters of SURVEY.md section 8c: exact in f32, so the f32 evaluation order of prediction.rs:190-204 cannot matter
    let vp: Vec<[f32; 6]> = vec![[0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625]; 3];
    let wp: Vec<[f32; 6]> = vec![[1.0, 0.5, 0.25, 0.25, 0.125, 0.125]; 3];
    let mut z = Npz::new();
    z.scalar_i64("width", w as i64);
    z.scalar_i64("height", h as i64);
    z.scalar_i64("channels", 3);
    z.add("qmatrix", "<i4", &[32], bytes_i32(&[1i32; 32])); // get_quantization_matrix (stages/quantization.rs:3-5)

    let raw = WaveletImage::from_raster(raster(w, h));
    let centers = canonical_centers(&raw.fractal_lattice);
    let f = centers.len();
    z.add("centers", "<i4", &[f, 2], bytes_i32(&centers.iter().flat_map(|c| [c.re, c.im]).collect::<Vec<i32>>()));
    z.add("coefs_raw", "<i4", &[3, f, NODES], bytes_i32(&coefficient_array(&raw, &centers)));
    let mut quantised = quantization::encode(raw).expect("quantization::encode");
    z.add("coefs", "<i4", &[3, f, NODES], bytes_i32(&coefficient_array(&quantised, &centers)));
    for ch in 0..3 {
        let (b, p, hist, oob) = predict_given(&quantised, &centers, ch, &vp, &wp);
        z.add(&format!("value_params_{}", ch), "<f4", &[3, 6], bytes_f32(&params_flat(&vp)));
        z.add(&format!("width_params_{}", ch), "<f4", &[3, 6], bytes_f32(&params_flat(&wp)));
        z.add(&format!("bucket_{}", ch), "|u1", &[f, NODES], b);
        z.add(&format!("prediction_{}", ch), "<i4", &[f, NODES], bytes_i32(&p));
        z.add(&format!("hist_{}", ch), "<u4", &[CONTEXT_AMOUNT, ALPHABET], bytes_u32(&hist));
        z.scalar_u64(&format!("oob_{}", ch), oob);
    }
    // the reference's own scan with its own fit: parameters as it transmits them, and the predictors it leaves in the lattice
    let mut opts = EncoderOpts::default();
    prediction::encode(&mut quantised, &mut opts).expect("prediction::encode");
    for ch in 0..3 {
        let (b, p) = predictor_arrays(&quantised, &centers, ch);
        z.add(&format!("fit_value_params_{}", ch), "<f4", &[3, 6], bytes_f32(&params_flat(&opts.value_prediction_params[ch])));
        z.add(&format!("fit_width_params_{}", ch), "<f4", &[3, 6], bytes_f32(&params_flat(&opts.width_prediction_params[ch])));
        z.add(&format!("fit_bucket_{}", ch), "|u1", &[f, NODES], b);
        z.add(&format!("fit_prediction_{}", ch), "<i4", &[f, NODES], bytes_i32(&p));
    }
    let decoded = RasterImage::from_wavelet(quantised); // extract_values + set_pixel clamp (:358-381, images.rs:103-111); quantization::decode is the identity for [1; 32]
    z.add("decoded", "|u1", &[(w * h * 3) as usize], decoded.data);
    z.write(&dir.join(format!("ref_kat_{}x{}_rgb.npz", w, h)));
}

#[test]
fn dump_golden() {
    let dir = std::path::PathBuf::from(std::env::var("FRI_GOLDEN_DIR").unwrap_or_else(|_| String::from(".")));
    for (w, h) in [(10u32, 10u32), (64, 48), (100, 37)] {
        dump(w, h, &dir);
        println!("wrote ref_kat_{}x{}_rgb.npz into {}", w, h, dir.display());
    }
}
