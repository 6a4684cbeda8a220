#!/bin/bash
# build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Compiles the UNMODIFIED reference sources where they lie under
# /root/reference (never copied into this repo) with amdflang into
# oracle/_ref/ (git-ignored, travels to the GPU box with gpurun):
#
#   kernels [NDIM]   -> oracle/_ref/libref_kernels{NDIM}d.so
#                       the reference's unsplit/ctoprim/uslope/trace*/cmpflxm/
#                       riemann_*/cmpdt behind oracle/ref_shim.f90 (bind(C))
#   ramses  [NDIM] [mpi|serial] [PATCHDIR]
#                    -> oracle/_ref/ramses{NDIM}d[_mpi][_<patchname>]
#                       the whole reference program (own recipe: the ordered
#                       object list below; the reference's Makefile is not run)
#
# The x86-64 baseline target has no FMA, so the objects are bit-reproducible
# against the reference's own gfortran-made goldens.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${RAMSES_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
F90="${F90:-amdflang}"
OPT="${REF_OPT:--O2}"
NVECTOR="${NVECTOR:-32}"

if [ ! -d "$REF/hydro" ]; then
  echo "build_ref.sh: reference tree not found at $REF (prebuilt oracle/_ref is used as is)" >&2
  exit 0
fi
mkdir -p "$OUT"

defines() { # ndim [nvar]   (REF_SOLVER=mhd: the reference's SOLVER=mhd build, NVAR=8)
  local ndim=$1 nvar=${2:-$(( $1 + 2 ))}
  if [ "${REF_SOLVER:-hydro}" = mhd ]; then
    echo "-cpp -DNVECTOR=$NVECTOR -DNDIM=$ndim -DNPRE=8 -DNENER=0 -DNVAR=${2:-8} -DSOLVERmhd"
  else
    echo "-cpp -DNVECTOR=$NVECTOR -DNDIM=$ndim -DNPRE=8 -DNENER=0 -DNVAR=$nvar -DSOLVERhydro"
  fi
}

build_kernels() { # ndim [nvar]: nvar>ndim+2 adds passive scalars (tag _vN)
  local ndim=$1 nvar=${2:-$(( $1 + 2 ))}
  local tag="${ndim}d"
  [ "$nvar" != "$((ndim+2))" ] && tag="${ndim}d_v${nvar}"
  local obj="$OUT/obj_kernels${tag}"
  mkdir -p "$obj"
  local flags="$(defines $ndim $nvar) -DWITHOUTMPI -fPIC $OPT -module-dir $obj -I$obj"
  local srcs=(amr/amr_parameters.f90 amr/amr_commons.f90 hydro/hydro_parameters.f90 hydro/hydro_commons.f90
              poisson/poisson_parameters.f90 poisson/poisson_commons.f90
              hydro/umuscl.f90 hydro/uplmde.f90 hydro/godunov_utils.f90 hydro/interpol_hydro.f90)
  local objs=()
  # hydro_commons needs amr_commons only for its module 'const'? compile the
  # minimal chain and let the compiler tell us if something is missing.
  for s in "${srcs[@]}"; do
    local o="$obj/$(basename "${s%.f90}").o"
    $F90 $flags -c "$REF/$s" -o "$o"
    objs+=("$o")
  done
  $F90 $flags -c "$HERE/ref_shim.f90" -o "$obj/ref_shim.o"
  $F90 -shared -o "$OUT/libref_kernels${tag}.so" "${objs[@]}" "$obj/ref_shim.o"
  echo "built $OUT/libref_kernels${tag}.so"
}

build_kernels_mhd() { # the MHD solver's mag_unsplit (NDIM=3, NVAR=8) behind oracle/ref_shim_mhd.f90
  local obj="$OUT/obj_kernels3d_mhd"
  mkdir -p "$obj"
  local flags="-cpp -DNVECTOR=$NVECTOR -DNDIM=3 -DNPRE=8 -DNENER=0 -DNVAR=8 -DSOLVERmhd -DWITHOUTMPI -fPIC $OPT -module-dir $obj -I$obj"
  local srcs=(amr/amr_parameters.f90 amr/amr_commons.f90 mhd/hydro_parameters.f90 hydro/hydro_commons.f90
              poisson/poisson_parameters.f90 poisson/poisson_commons.f90 mhd/umuscl.f90 mhd/godunov_utils.f90)
  local objs=()
  for s in "${srcs[@]}"; do
    local o="$obj/$(basename "${s%.f90}").o"
    $F90 $flags -c "$REF/$s" -o "$o"
    objs+=("$o")
  done
  $F90 $flags -c "$HERE/ref_shim_mhd.f90" -o "$obj/ref_shim_mhd.o"
  $F90 -shared -o "$OUT/libref_kernels3d_mhd.so" "${objs[@]}" "$obj/ref_shim_mhd.o"
  echo "built $OUT/libref_kernels3d_mhd.so"
}

# Ordered object list of the full program (module files first), resolved
# through the same search order the reference uses: PATCH, hydro, pm, poisson,
# amr, io.
MODSRC="mpi_mod amr_parameters amr_commons random pm_parameters sink_feedback_parameters
 pm_commons poisson_parameters dump_utils constants file_module
 poisson_commons hydro_parameters hydro_commons cooling_module bisection sparse_mat
 clfind_commons gadgetreadfile write_makefile write_patch write_gitinfo sink_sn_feedback"
AMRSRC="read_params init_amr init_time init_refine tracer_utils adaptive_loop amr_step
 update_time output_amr flag_utils physical_boundaries virtual_boundaries refine_utils
 nbors_utils hilbert load_balance title sort cooling_fine eos units light_cone movie
 memory end"
PMSRC="init_part output_part rho_fine synchro_fine move_fine newdt_fine particle_tree
 add_list remove_list star_formation sink_particle feedback clump_finder clump_merger
 output_clump flag_formation_sites init_sink output_sink unbinding merger_tree
 move_tracer init_tracer read_sink_feedback_params sink_rt_feedback stellar_particle
 init_stellar output_stellar"
POISSONSRC="init_poisson phi_fine_cg interpol_phi force_fine multigrid_coarse
 multigrid_fine_commons multigrid_fine_fine multigrid_fine_coarse gravana
 boundary_potential rho_ana output_poisson"
HYDROSRC="init_hydro init_flow_fine write_screen output_hydro courant_fine godunov_fine
 uplmde umuscl interpol_hydro godunov_utils condinit hydro_flag hydro_boundary boundana
 read_hydro_params synchro_hydro_fine cooling_module_ism"

build_ramses() {
  local ndim=$1 mode=${2:-serial} patch=${3:-}
  local tag="ramses${ndim}d"
  [ "$mode" = mpi ] && tag="${tag}_mpi"
  [ -n "$patch" ] && tag="${tag}_$(basename "$patch")"
  # extra cpp defines of the reference itself, e.g. REF_DEFS=-DOUTPUT_PARTICLE_DENSITY
  # (backup_poisson then also writes rho) with REF_TAG=rho
  [ -n "${REF_TAG:-}" ] && tag="${tag}_${REF_TAG}"
  local obj="$OUT/obj_$tag" gen="$OUT/gen_$tag"
  mkdir -p "$obj" "$gen"
  local flags="$(defines $ndim ${REF_NVAR:-}) ${REF_DEFS:-} $OPT -module-dir $obj -I$obj -I$REF"   # REF_NVAR: passive scalars
  local libs=""
  if [ "$mode" = mpi ]; then
    flags="$flags -DMPI_OLD -I/opt/conda/include"
    libs="-L/opt/conda/lib -lmpifort -lmpi -Wl,-rpath,/opt/conda/lib"
  else
    flags="$flags -DWITHOUTMPI"
  fi
  local extra_objs=""
  if [ -n "$patch" ] && [ -f "$patch/build_flags.sh" ]; then
    # a patch may add compile flags / extra objects / link libraries
    # shellcheck disable=SC1090
    source "$patch/build_flags.sh"
    flags="$flags ${PATCH_FFLAGS:-}"
    libs="$libs ${PATCH_LIBS:-}"
    extra_objs="${PATCH_EXTRA_SRC:-}"
  fi
  if [ -n "$patch" ] && [ -x "$patch/prepare.sh" ]; then
    # sources a patch derives from the reference tree (generated next to the other stubs, included by its shims)
    "$patch/prepare.sh" "$REF" "$gen"
    flags="$flags -I$gen"
  fi
  # generated stubs the reference's Makefile would create with shell scripts
  cat > "$gen/write_makefile.f90" <<'EOF'
subroutine output_makefile(filename)
  character(LEN=80)::filename
  open(unit=11,file=TRIM(filename),form='formatted')
  write(11,'(A)')'built by oracle/build_ref.sh'
  close(11)
end subroutine output_makefile
EOF
  cat > "$gen/write_patch.f90" <<'EOF'
subroutine output_patch(filename)
  character(LEN=80)::filename
  open(unit=11,file=TRIM(filename),form='formatted')
  write(11,'(A)')'see oracle/build_ref.sh'
  close(11)
end subroutine output_patch
EOF
  cat > "$gen/write_gitinfo.f90" <<'EOF'
subroutine write_gitinfo
  use amr_commons, ONLY:builddate,patchdir,gitrepo,gitbranch,githash
  builddate = 'oracle/build_ref.sh'
  patchdir  = 'see build tag'
  gitrepo   = 'tatary/ramses'
  gitbranch = 'reference'
  githash   = 'reference'
end subroutine write_gitinfo
EOF
  if [ "$ndim" != 3 ]; then
    # flang rejects a rank-mismatched assignment in dead NDIM<3 code of
    # pm/sink_sn_feedback.f90; fix it on the fly.
    sed 's/xx(i, :) = xg(ind_grid(i), :) + xc(ind, :)/xx(i, 1:ndim) = xg(ind_grid(i), 1:ndim) + xc(ind, 1:ndim)/' \
        "$REF/pm/sink_sn_feedback.f90" > "$gen/sink_sn_feedback.f90"
  fi
  find_src() { # name -> path
    local n=$1
    local solver_dir=""
    [ "${REF_SOLVER:-hydro}" = mhd ] && solver_dir="$REF/mhd"      # bin/Makefile: VPATH puts ../mhd before ../hydro
    for d in "$patch" "$gen" "$solver_dir" "$REF/hydro" "$REF/pm" "$REF/poisson" "$REF/amr" "$REF/io"; do
      [ -n "$d" ] || continue
      for ext in f90 F; do
        if [ -f "$d/$n.$ext" ]; then echo "$d/$n.$ext"; return; fi
      done
    done
    echo "MISSING:$n" >&2; return 1
  }
  local objs=()
  # a patch module that changed (the C ABI's interfaces, the shims' shared state) rebuilds everything that may `use` it:
  # an object older than the newest module source of the patch directory is stale whatever its own source says
  local newest_mod=""
  if [ -n "$patch" ]; then
    newest_mod=$(ls -t "$patch"/ramses_amd_*.f90 2>/dev/null | head -1 || true)      # (a patch directory without modules: dump_patch)
  fi
  for n in $MODSRC $extra_objs $AMRSRC $HYDROSRC $PMSRC $POISSONSRC ramses; do
    local src; src=$(find_src "$n")
    local o="$obj/$n.o"
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || { [ -n "$newest_mod" ] && [ "$newest_mod" -nt "$o" ]; } || [ "${FORCE:-0}" = 1 ]; then
      $F90 $flags -c "$src" -o "$o" 2> "$obj/$n.log" || { cat "$obj/$n.log"; exit 1; }
    fi
    objs+=("$o")
  done
  $F90 "${objs[@]}" -o "$OUT/$tag" $libs
  echo "built $OUT/$tag"
}

cmd=${1:-kernels}
case "$cmd" in
  kernels) build_kernels "${2:-3}" "${3:-}";;
  kernels_mhd) build_kernels_mhd;;
  ramses) build_ramses "${2:-3}" "${3:-serial}" "${4:-}";;
  all) # every artefact tests/ and bench.py look for; the programs are independent (own object and stub directories):
       # built side by side, REF_JOBS at a time (default: the number of cores)
       JOBS=${REF_JOBS:-$(nproc 2>/dev/null || echo 4)}
       fail=0
       bg() { # run "$@" in a subshell, at most JOBS at a time
         while [ "$(jobs -rp | wc -l)" -ge "$JOBS" ]; do wait -n || fail=1; done
         ( "$@" ) &
       }
       k() { build_kernels "$@"; }
       r() { build_ramses "$@"; }
       r_rho() { REF_DEFS=-DOUTPUT_PARTICLE_DENSITY REF_TAG=rho build_ramses 3 serial; }
       r_v7() { REF_NVAR=7 REF_TAG=v7 build_ramses 3 serial "$@"; }
       r_mhd() { REF_SOLVER=mhd REF_TAG=mhd build_ramses 3 serial "$@"; }
       bg k 3; bg k 1; bg k 2; bg k 3 7
       bg build_kernels_mhd
       bg r 3 serial; bg r 1 serial; bg r 2 serial
       bg r_rho
       if [ -d /opt/conda/include ] && [ -f /opt/conda/lib/libmpifort.so ]; then bg r 3 mpi; fi
       if [ -f "$HERE/../ramses_amd/lib/libramses_amd.so" ]; then
         bg r 3 serial "$HERE/../ramses_amd/patch"
         bg r 1 serial "$HERE/../ramses_amd/patch"     # the shims compile for NDIM=1,2 (A/B tests with RAMSES_AMD=0)
         bg r 2 serial "$HERE/../ramses_amd/patch"
         if [ -d /opt/conda/include ] && [ -f /opt/conda/lib/libmpifort.so ]; then
           bg r 3 mpi "$HERE/../ramses_amd/patch"
         fi
       fi
       bg r_mhd
       if [ -f "$HERE/../ramses_amd/lib/libramses_amd.so" ]; then bg r_mhd "$HERE/../ramses_amd/patch_mhd"; fi
       # SOLVER=mhd with MPI (the AMR levels of an MHD run on several ranks: tests/test_mhd_amr_gpu.py)
       r_mhd_mpi() { REF_SOLVER=mhd REF_TAG=mhd build_ramses 3 mpi "$@"; }
       if [ -d /opt/conda/include ] && [ -f /opt/conda/lib/libmpifort.so ]; then
         bg r_mhd_mpi
         if [ -f "$HERE/../ramses_amd/lib/libramses_amd.so" ]; then bg r_mhd_mpi "$HERE/../ramses_amd/patch_mhd"; fi
       fi
       bg r 3 serial "$HERE/dump_patch"
       bg r_v7
       if [ -f "$HERE/../ramses_amd/lib/libramses_amd.so" ]; then
         bg r_v7 "$HERE/../ramses_amd/patch"
       fi
       while [ "$(jobs -rp | wc -l)" -gt 0 ]; do wait -n || fail=1; done
       if [ "$fail" != 0 ]; then echo "build_ref.sh: a build failed" >&2; exit 1; fi;;
  *) echo "usage: $0 kernels [NDIM] | ramses [NDIM] [serial|mpi] [PATCHDIR] | all"; exit 2;;
esac
