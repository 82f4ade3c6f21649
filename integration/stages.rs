// stages.rs -- the replacement bodies a maintainer of pagmerek/frave puts behind libfri's private stage functions so that FRIEncoder::encode /
// FRIDecoder::decode (crates/libfri/src/encoder.rs:87-109, decoder.rs:47-59) run their hot path on an MI355X through hip_sys.rs / emit_sys.rs.
// UNVERIFIED SOURCE: the build image has no Rust toolchain. The calls, their argument meaning and their order are the ones of the C++ mirror
// frave_amd/host/libfri.cpp, which IS compiled and tested against the CPU oracle (tests/), through the same C ABI.
//
// What changes in the reference: stages/wavelet_transform.rs::encode, stages/quantization.rs::encode and stages/prediction.rs::encode lose their
// loops (A); or, shorter, FRIEncoder::encode calls the chain entry point and enters the state machine at EntropyEncoding (B); or, shortest,
// encode_bytes (C) lets the device also run the emitter's gather and feeds the existing rANS loop with 2-byte symbols.
// `opts.hip` is a small cache { ctx: *mut fri_hip_ctx, plans: HashMap<(u32, u32, u32), *mut fri_hip_plan> } added to EncoderOpts.

use crate::hip_sys::*;
use crate::emit_sys::*;

fn check(rc: std::os::raw::c_int) -> Result<(), String> {
    if rc == FRI_HIP_OK { Ok(()) } else { Err(unsafe { std::ffi::CStr::from_ptr(fri_hip_strerror(rc)) }.to_string_lossy().into_owned()) }
}

// ---- (A) stage by stage ------------------------------------------------------------------------------------------------------------------
// stages/wavelet_transform.rs -- replaces the extract_coefficients loop of WaveletImage::from_raster (:412-416) AND quantization::encode (:7-25)
pub fn wavelet_transform_encode(raster: RasterImage, opts: &EncoderOpts) -> Result<WaveletImage, String> {
    let (w, h, c) = (raster.metadata.width, raster.metadata.height, raster.metadata.colorspace.num_channels() as u32);
    let plan = opts.hip.plan(w, h, c)?; // cached per (w, h, c); wraps fri_hip_plan_create
    let f = unsafe { fri_hip_plan_num_cells(plan) } as usize;
    let mut coefs = vec![0i32; c as usize * f * 512];
    let q = quantization::get_quantization_matrix(); // [1; 32] today (stages/quantization.rs:3-5)
    check(unsafe { fri_hip_transform_quant(plan, raster.data.as_ptr(), q.as_ptr(), coefs.as_mut_ptr()) })?;
    let mut centers = vec![0i32; 2 * f];
    check(unsafe { fri_hip_plan_centers(plan, centers.as_mut_ptr()) })?;
    let mut lattice = HashMap::with_capacity(f);
    for k in 0..f {
        let center = Complex::new(centers[2 * k], centers[2 * k + 1]);
        let mut frac = Fractal::new(BASE_FRAC_DEPTH, center); // geometry only: image_positions / position_map
        for ch in 0..c as usize {
            let base = (ch * f + k) * 512;
            frac.coefficients[ch] = coefs[base..base + 512].iter().map(|&v| if v == FRI_HIP_NONE { None } else { Some(v) }).collect();
        }
        lattice.insert(center, frac); // only retained cells exist: the retain() at :415 is implied
    }
    let gpm = WaveletImage::get_global_position_map(&lattice);
    let sorted = WaveletImage::sort_lattice(&lattice, &gpm, h, w);
    Ok(WaveletImage { metadata: raster.metadata, fractal_lattice: lattice, global_position_map: gpm, sorted_lattice: sorted })
}
// quantization::encode becomes the identity (already applied on the device with the same matrix).

// stages/prediction.rs -- replaces optimize_parameters (:232-235) and the three scan loops (:241-298) for ALL channels: one upload, fit and scan on the device
pub fn prediction_encode(image: &mut WaveletImage, opts: &mut EncoderOpts, coefs: &[i32], centers: &[i32]) -> Result<[Vec<AnsContext>; 3], String> {
    let c = image.metadata.colorspace.num_channels();
    let plan = opts.hip.plan(image.metadata.width, image.metadata.height, c as u32)?;
    let f = centers.len() / 2;
    let (mut vp, mut wp) = (vec![0f32; c * 18], vec![0f32; c * 18]); // fitted [channels][3][6] parameters come back here
    let (mut bucket, mut pred) = (vec![0u8; c * f * 512], vec![0i32; c * f * 512]);
    let (mut hist, mut oob) = (vec![0u32; c * 10 * 1024], vec![0u64; c]);
    check(unsafe { fri_hip_predict_image(plan, coefs.as_ptr(), 1, vp.as_mut_ptr(), wp.as_mut_ptr(), bucket.as_mut_ptr(), pred.as_mut_ptr(), hist.as_mut_ptr(), oob.as_mut_ptr()) })?;
    if oob.iter().any(|&n| n != 0) { return Err("symbol outside the 1024-entry alphabet".into()); } // the reference panics here (entropy_coding.rs:99)
    let mut contexts: [Vec<AnsContext>; 3] = Default::default();
    for ch in 0..c {
        opts.value_prediction_params[ch] = to_groups(&vp[ch * 18..(ch + 1) * 18]); // transmitted in the PRD segment (serialize.rs:78-91)
        opts.width_prediction_params[ch] = to_groups(&wp[ch * 18..(ch + 1) * 18]);
        for k in 0..f {
            let frac = image.fractal_lattice.get_mut(&Complex::new(centers[2 * k], centers[2 * k + 1])).unwrap();
            for i in 0..512 { frac.parameter_predictors[ch][i] = (bucket[(ch * f + k) * 512 + i] as usize, pred[(ch * f + k) * 512 + i]); }
        }
        contexts[ch] = (0..10).map(|b| {
            let mut ctx = AnsContext::new();
            ctx.freqs.copy_from_slice(&hist[(ch * 10 + b) * 1024..(ch * 10 + b + 1) * 1024]);
            ctx // then max_freq_bits / finalize_context exactly as prediction.rs:302-318
        }).collect();
    }
    Ok(contexts)
}

// ---- (B) FRIEncoder::encode's three device stages as ONE call (encoder.rs:19-38) --------------------------------------------------------------
// The pixels go up once, the coefficients stay in device memory between the stages (the fit's 6 x 6 solves run on the device too), every output
// comes down once; the state machine then enters at EncoderStage::EntropyEncoding with the WaveletImage rebuilt from the flat arrays as in (A).
pub fn device_stages(raster: &RasterImage, opts: &mut EncoderOpts) -> Result<DeviceOutputs, String> {
    let (w, h, c) = (raster.metadata.width, raster.metadata.height, raster.metadata.colorspace.num_channels());
    let plan = opts.hip.plan(w, h, c as u32)?;
    let f = unsafe { fri_hip_plan_num_cells(plan) } as usize;
    let q = quantization::get_quantization_matrix();
    let mut o = DeviceOutputs::with_sizes(c, f);
    check(unsafe { fri_hip_encode_image(plan, raster.data.as_ptr(), q.as_ptr(), 1, o.vp.as_mut_ptr(), o.wp.as_mut_ptr(), o.coefs.as_mut_ptr(),
                                        o.bucket.as_mut_ptr(), o.pred.as_mut_ptr(), o.hist.as_mut_ptr(), o.oob.as_mut_ptr()) })?;
    Ok(o)
}

// ---- (C) encode_bytes with the emitter's gather on the device --------------------------------------------------------------------------------
// sort_lattice's order (wavelet_transform.rs:657-705) is geometry: fri_emit_stream_order builds it once per plan (None nodes taken out) and the plan
// keeps it; fri_hip_encode_image_symbols then returns, per channel, bucket << 10 | symbol in that order (2 bytes per symbol instead of 9 bytes per
// node), which is exactly what the loop of entropy_coding::encode (:285-336) feeds to rans' encoder.put - in reverse - with contexts finalised
// from `hist` (prediction.rs:302-318). serialize::encode is unchanged.
pub fn encode_symbols_on_device(raster: &RasterImage, opts: &mut EncoderOpts) -> Result<(Vec<u16>, Vec<u32>, usize), String> {
    let (w, h, c) = (raster.metadata.width, raster.metadata.height, raster.metadata.colorspace.num_channels());
    let plan = opts.hip.plan(w, h, c as u32)?;
    let f = unsafe { fri_hip_plan_num_cells(plan) };
    let n = unsafe { fri_hip_plan_num_some(plan) } as usize;
    if !opts.hip.has_stream_order(plan) { // once per plan
        let (mut centers, mut mask, mut order, mut n_order) = (vec![0i32; 2 * f as usize], vec![0u32; 16 * f as usize], vec![0u32; 512 * f as usize], 0u64);
        check(unsafe { fri_hip_plan_centers(plan, centers.as_mut_ptr()) })?;
        check(unsafe { fri_hip_plan_valid_mask(plan, mask.as_mut_ptr()) })?;
        if unsafe { fri_emit_stream_order(centers.as_ptr(), f, mask.as_ptr(), order.as_mut_ptr(), &mut n_order) } != 0 { return Err("stream order".into()); }
        check(unsafe { fri_hip_plan_set_stream_order(plan, order.as_ptr(), n_order) })?;
        opts.hip.mark_stream_order(plan);
    }
    let q = quantization::get_quantization_matrix();
    let (mut vp, mut wp) = (vec![0f32; c * 18], vec![0f32; c * 18]);
    let (mut symbols, mut hist, mut oob) = (vec![0u16; c * n], vec![0u32; c * 10 * 1024], vec![0u64; c]);
    check(unsafe { fri_hip_encode_image_symbols(plan, raster.data.as_ptr(), q.as_ptr(), 1, vp.as_mut_ptr(), wp.as_mut_ptr(), symbols.as_mut_ptr(), hist.as_mut_ptr(), oob.as_mut_ptr()) })?;
    if oob.iter().any(|&k| k != 0) { return Err("symbol outside the 1024-entry alphabet".into()); }
    store_params(opts, &vp, &wp);
    Ok((symbols, hist, n)) // channel ch: symbols[ch * n..(ch + 1) * n]; bucket = s >> 10, symbol = s & 1023
}

// ---- batches (crates/fri-cli/src/commands/bench.rs:15-120: the loop over the images of a directory) ---------------------------------------------
// fri_hip_encode_image_batch (one GPU, three streams: uploads, kernels and downloads of consecutive images overlap) and
// fri_hip_multi_encode_image (image i on GPU i mod N, one host thread per GPU, no data between GPUs) take arrays of per-image pointers.

// ---- decoder (decoder.rs:17-41) ----------------------------------------------------------------------------------------------------------------
// quantization::decode + wavelet_transform::decode collapse into one fri_hip_inverse_transform call on the flattened coefficients
// (None -> FRI_HIP_NONE). The ABI reproduces the reference's *dividing* dequantiser (quantization.rs:37) bit for bit; with today's all-ones
// matrix that is the identity.
pub fn wavelet_transform_decode(image: &WaveletImage, coefs: &[i32], opts: &EncoderOpts) -> Result<RasterImage, String> {
    let (w, h, c) = (image.metadata.width, image.metadata.height, image.metadata.colorspace.num_channels() as u32);
    let plan = opts.hip.plan(w, h, c)?;
    let mut data = vec![0u8; unsafe { fri_hip_plan_pixel_bytes(plan) }];
    let q = quantization::get_quantization_matrix();
    check(unsafe { fri_hip_inverse_transform(plan, coefs.as_ptr(), q.as_ptr(), data.as_mut_ptr()) })?;
    Ok(RasterImage { metadata: image.metadata.clone(), data })
}
