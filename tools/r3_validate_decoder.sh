#!/bin/bash
# after a change to the decoder: the randomized parity run (every other file decoded here and compared with Pillow), the sweep,
# and 150 decodes of the full-size file compared pixel for pixel
cd $GRAFT_REPO_ROOT
{
echo "== library hash $(python -c 'import nvjpeg_imagecompressor_amd as m; print(m.library_source_hash())')"
echo "== tools/fuzz_parity.py 3000 31337"; timeout -k 10 500 python tools/fuzz_parity.py 3000 31337 | tail -2 || exit 1
echo "== tests/sweep_random_parity.py 120 2718 6"; timeout -k 10 300 python tests/sweep_random_parity.py 120 2718 6 | tail -1 || exit 1
echo "== tools/decode_hammer.py 150"; timeout -k 10 300 python tools/decode_hammer.py 150 2>/dev/null | tail -3 || exit 1
} 2>&1 | tee gpurun_out/validate_decoder.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a gpurun_out/validate_decoder.txt
