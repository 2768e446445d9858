#!/bin/bash
# tests/cmp_ref.sh <fasta> <dwgsim options...>
# Dev-container check: run the unmodified reference (oracle/_ref/dwgsim) and the oracle in mode A
# (sequential drand48) with the same options and compare all five outputs byte-for-byte
# (FASTQ after gunzip, as the reference's own testdata/test.sh:21-26 does).
set -u
FA=$1; shift
T=$(mktemp -d /tmp/cmpref.XXXXXX)
/root/repo/oracle/_ref/dwgsim "$@" "$FA" $T/ref > $T/ref.log 2>&1; rc1=$?
/root/repo/oracle/build/dwgsim_oracle --rng drand48 --verbose "$@" "$FA" $T/ora > $T/ora.log 2>&1; rc2=$?
ok=1
for f in bfast.fastq bwa.read1.fastq bwa.read2.fastq; do
  if [ -f $T/ref.$f.gz ]; then
    if ! cmp -s <(zcat $T/ref.$f.gz) $T/ora.$f; then echo "DIFF $f"; ok=0; fi
  elif [ -f $T/ora.$f ]; then echo "EXTRA $f"; ok=0; fi
done
for f in mutations.txt mutations.vcf; do
  if [ -f $T/ref.$f ]; then
    if ! cmp -s $T/ref.$f $T/ora.$f; then echo "DIFF $f"; ok=0; fi
  fi
done
n=$(zcat $T/ref.bwa.read1.fastq.gz 2>/dev/null | wc -l); m=$(wc -l < $T/ref.mutations.txt 2>/dev/null)
if [ $ok = 1 ]; then echo "OK   rc=$rc1/$rc2 lines1=$n muts=$m :: $* :: $(tail -1 $T/ora.log | cut -c1-150)"; rm -rf $T; else echo "FAIL rc=$rc1/$rc2 :: $* (kept $T)"; fi
