#!/bin/bash
# round 6, GPU call 24: the tables in the order of the file's own scores (tm_vocab_load's default) against the file's order and against a sample's order
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe24; mkdir -p $OUT
for cfg in englishcode-32000-consistent englishcode-100256-clean code-4096-balanced-nocapcode; do
  for mode in file score sample; do
    case $mode in
      file) env TM_LAYOUT=file python tools/k1_time.py --config $cfg --mbytes 512 --reps 5 2>&1 | grep -v Warn | tail -1 ;;
      score) python tools/k1_time.py --config $cfg --mbytes 512 --reps 5 2>&1 | grep -v Warn | tail -1 ;;
      sample) env TM_K1_TUNE_MIB=16 python tools/k1_time.py --config $cfg --mbytes 512 --reps 5 2>&1 | grep -v Warn | tail -1 ;;
    esac
  done
done | tee $OUT/k1_layouts.txt
for mode in file score; do
  [ $mode = file ] && export TM_LAYOUT=file || unset TM_LAYOUT
  python tools/k1_time.py --config candidates-65536 --score --mbytes 512 --reps 5 2>&1 | grep -v Warn | tail -1
done | tee -a $OUT/k1_layouts.txt
