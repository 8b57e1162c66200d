# round 3, GPU call 3: where does a Householder step of the batched eigen-solver spend its time (variants, timing only)
OUT=gpurun_out/r3c; mkdir -p $OUT
for T in 256 512 900; do
for v in "0 0" "0 1" "0 10" "0 11" "0 13" "4 0" "4 1" "4 10" "4 12" "8 0" "8 1" "8 10" "8 12"; do
  set -- $v
  CGE_EIG_FORM=$1 CGE_EIG_DIAG=$2 timeout -k 5 120 python profiles/eig_stage_timing.py $T 128 >> $OUT/eig_variants.txt 2>> $OUT/eig_variants.err || { echo "variant $v failed"; tail -3 $OUT/eig_variants.err; }
done; done
cat $OUT/eig_variants.txt
