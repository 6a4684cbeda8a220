# sourced by oracle/build_ref.sh (which compiles the reference tree without
# running its Makefile): the same additions as ./Makefile
_here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PATCH_FFLAGS=""
PATCH_EXTRA_SRC="ramses_amd_cabi ramses_amd_iface"
PATCH_LIBS="-L$_here/../lib -lramses_amd -Wl,-rpath,$_here/../lib -Wl,-rpath,/opt/rocm/lib"
