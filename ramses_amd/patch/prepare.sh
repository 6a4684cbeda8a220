#!/bin/bash
# prepare.sh REF GEN -- derived sources of the ramses_amd patch, generated from the reference tree where it lies
# (nothing of it is kept in this repository):
#   GEN/multigrid_fine_commons_ref.f90 = poisson/multigrid_fine_commons.f90 with the DEFINITIONS of make_virtual_mg_dp and
#   make_reverse_mg_dp renamed to *_reference.  Their callers (multigrid_fine, recursive_multigrid_coarse) sit in the same
#   file, so the preprocessor rename the other shims use would rename the calls as well; this way the calls reach the
#   routines of the same name in ramses_amd/patch/multigrid_fine_commons.f90.
set -e
REF=$1; GEN=$2
mkdir -p "$GEN"
sed -E 's/(subroutine[ ]+)(make_virtual_mg_dp|make_reverse_mg_dp)\b/\1\2_reference/I' \
    "$REF/poisson/multigrid_fine_commons.f90" > "$GEN/multigrid_fine_commons_ref.f90.tmp"
# (keep the time stamp unless the content changed: the build compiles what is newer than its object)
if ! cmp -s "$GEN/multigrid_fine_commons_ref.f90.tmp" "$GEN/multigrid_fine_commons_ref.f90" 2>/dev/null; then
  mv "$GEN/multigrid_fine_commons_ref.f90.tmp" "$GEN/multigrid_fine_commons_ref.f90"
else
  rm -f "$GEN/multigrid_fine_commons_ref.f90.tmp"
fi
grep -c "subroutine make_virtual_mg_dp_reference\|subroutine make_reverse_mg_dp_reference" "$GEN/multigrid_fine_commons_ref.f90" | grep -qx 4
