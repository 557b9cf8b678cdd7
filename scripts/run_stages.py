#!/usr/bin/env python3
"""End-to-end run of the reference's test pipeline (stages 4 -> 5 -> 6b/6c of
`LDPC_128/Training and Testing recipe.txt`) on the MI355X path, with the reference's file layout:

  data/snr<lo>-<hi>dB/test-nonzero<snr>dB-Awgn.tfrecord            (stage 4, Testing_data_gen_128/Main_test.py)
  data/.../<type>/<T>th/<snr>dB/ldpc-nonzero-retest.tfrecord        (stage 5, Ldpc_128_testing/ldpc_128_testing.py)
  log/FER-<type>-<T>th.txt, log/PB-OSD-order-<p>.txt, log/FS-OSD-order-<p>.txt, log/CNV-OSD-order-<p>.txt

    python scripts/run_stages.py --out /tmp/ldpc_run --snr 2.0 3.0 3 --frames 20000 --iters 10 --order 2

Every decode runs through libldpcosd.so; files are written/read by the TensorFlow-free TFRecord codec.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from short_ldpc_decoding_osd_amd import Code, data_generating, fs_testing, ms_test, pb_testing, read_TFdata  # noqa: E402
from short_ldpc_decoding_osd_amd import globalmap as GL  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="./ldpc_run")
    ap.add_argument("--snr", nargs=3, default=["2.0", "3.0", "3"], metavar=("LO", "HI", "NUM"))
    ap.add_argument("--frames", type=int, default=20000, help="frames per SNR point")
    ap.add_argument("--batch", type=int, default=5000, help="unit batch of the NMS stage")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--weight", type=float, default=-0.048, help="stored (pre-softplus) NMS-1 weight, or use --values")
    ap.add_argument("--values", default=None, help="par/values.txt written by the training stage")
    ap.add_argument("--seed", type=int, default=20241020)
    args = ap.parse_args()

    os.makedirs(args.out, exist_ok=True)
    os.chdir(args.out)
    lo, hi, num = float(args.snr[0]), float(args.snr[1]), int(args.snr[2])
    code = Code()
    for k, v in dict(code_parameters=code, num_iterations=args.iters, selected_decoder_type='NMS-1',
                     ALL_ZEROS_CODEWORD_TESTING=False, order_limit=args.order, termination_num_threshlod=100,
                     d_min=14, tau_psc=30).items():
        GL.set_map(k, v)
    model = ms_test.Decoding_model()
    if args.values:
        from short_ldpc_decoding_osd_amd import weights
        print("loaded weights of step", weights.load_values_txt(model, args.values))
    else:
        model.set_check_weight(args.weight)
    data_dir = f'./data/snr{lo}-{hi}dB/'
    os.makedirs(data_dir, exist_ok=True)
    os.makedirs('./log', exist_ok=True)
    log_filename = f'./log/FER-NMS-1-{args.iters}th.txt'
    rng = np.random.default_rng(args.seed)
    FER_list = []
    for SNR in np.linspace(lo, hi, num):
        snr = round(float(SNR), 2)
        # ---- stage 4: test set
        test_file = f'{data_dir}test-nonzero{snr}dB-Awgn.tfrecord'
        y, labels = data_generating.testing_data_generating(code, SNR, args.frames, rng=rng)
        data_generating.make_tfrecord((y.astype(np.float32), labels), test_file)
        # ---- stage 5: NMS, FER log, retest file
        buffer_inputs, buffer_labels, tot_fer, tot_ber, und, batches = [], [], 0.0, 0.0, 0, 0
        for inputs in read_TFdata.data_handler(128, test_file, args.batch).as_numpy_iterator():
            fer, ber, undetected, buf = model(inputs[0], inputs[1])
            buffer_inputs.append(buf[0]); buffer_labels.append(buf[1])
            tot_fer += fer; tot_ber += ber; und += undetected; batches += 1
            if tot_fer > 40000 / args.batch:          # decoding_threshold, ldpc_128_testing.py:36,130
                break
        fer_nms = tot_fer / batches
        FER_list.append((snr, round(fer_nms, 5)))
        with open(log_filename, 'a+') as f:
            f.write('\nFor %.1fdB summary:\n' % snr)
            f.write("FER %.4f, BER %.4f,UFER %.6f" % (fer_nms, tot_ber / batches, und / (batches * args.batch)) + '\n')
        out_dir = f'{data_dir}NMS-1/{args.iters}th/{snr}dB/'
        os.makedirs(out_dir, exist_ok=True)
        retest = out_dir + 'ldpc-nonzero-retest.tfrecord'
        ms_test.save_decoded_data(model.postprocess_failure_cases((buffer_inputs, buffer_labels)), retest, SNR,
                                  log_filename, args.iters + 1)
        # ---- stage 6: OSD on the failures (conventional, PB, FS)
        ds = read_TFdata.data_handler(128, retest, args.iters + 1)
        GL.set_map('miracle_view', False)
        GL.set_map('convention_osd', True); GL.set_map('pb_osd', False); GL.set_map('fs_osd', False)
        cnv = pb_testing.pb_osd(snr, ds)['convention_osd']
        GL.set_map('convention_osd', False); GL.set_map('pb_osd', True)
        pb = pb_testing.pb_osd(snr, ds)['pb_osd']
        GL.set_map('pb_osd', False); GL.set_map('fs_osd', True)
        fs = fs_testing.fs_osd(snr, 0.1, ds)['fs_osd']
        print(f"== {snr} dB: FER_NMS {fer_nms:.4f} | OSD-{args.order} fail|NMS-fail: conventional {cnv['FER']:.4f} "
              f"PB {pb['FER']:.4f} ({pb['average_teps']:.1f} TEPs) FS {fs['FER']:.4f} ({fs['average_teps']:.1f} TEPs) "
              f"| end-to-end (product, recipe.txt:18): {fer_nms * cnv['FER']:.5f}")
    with open(log_filename, 'a+') as f:
        f.write(f"FER_list:{FER_list}")
    print(f"FER_list:{FER_list}")


if __name__ == "__main__":
    main()
