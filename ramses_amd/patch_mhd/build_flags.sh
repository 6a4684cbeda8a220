# sourced by oracle/build_ref.sh for a SOLVER=mhd build of the reference with this patch directory
_here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PATCH_FFLAGS=""
PATCH_EXTRA_SRC="ramses_amd_mhd_iface"
PATCH_LIBS="-L$_here/../lib -lramses_amd -Wl,-rpath,$_here/../lib -Wl,-rpath,/opt/rocm/lib"
