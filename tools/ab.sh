# same-box A/B of two builds of the library: abtmp/A.so and abtmp/B.so are swapped in turn (box-to-box variance is ~3 %,
# larger than most of the differences worth measuring); AB_CMD is the command whose output is compared
cd $GRAFT_REPO_ROOT
for v in A B A B; do
  cp abtmp/$v.so pyflwdir_amd/libpfd_hip.so
  echo "== $v"
  bash -c "${AB_CMD:-python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -c 400}"
done
cp abtmp/B.so pyflwdir_amd/libpfd_hip.so
