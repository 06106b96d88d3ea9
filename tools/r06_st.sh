mkdir -p gpurun_out/r06_st; export TMPDIR=/tmp
for ov in "" "fuse_bconv_max_in=15"; do
python3 tools/stage_times_batch.py 8 hmult config_4.cfg 28 28 28 $ov
python3 tools/stage_times_batch.py 8 hmult config_4_N15.cfg 28 28 28 $ov
python3 tools/stage_times_batch.py 8 hmult config_4.cfg 28 16 28 $ov
done
