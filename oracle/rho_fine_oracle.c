/* oracle/rho_fine_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of rho_fine's hydro deposit on one level of a periodic nx=ny=nz=1 box (3-D, no
 * particles, every cell of the level a leaf), in the reference's loop order:
 *   rho_fine              pm/rho_fine.f90:5-226     (multipole reset, level loop, rho_tot :176-183)
 *   multipole_fine        :666-820                  (mass and mass*position of every leaf cell into
 *                                                    unew(:,1:4), the reference's scratch)
 *   cic_from_multipole    :825-891, cic_cell :896-1142  (the mass of a cell is CIC-deposited at its
 *                                                    centre of mass (m*x)/m: not the cell density bit
 *                                                    for bit once rounding moves that point)
 *   get3cubefather        amr/nbors_utils.f90:5-194 (as x,y,z steps through son(nbor))
 * Pinned by tests/test_rho_fine_oracle.py against dumps of the unmodified reference
 * (oracle/dump_patch/rho_fine.f90 -> tests/golden/rho_fine_ref.npz).
 *
 * Arrays are the reference's: cell = ncoarse + ind*ngridmax + igrid (1-based igrid, ind 0..7),
 * xg = (3, ngridmax), nbor = (6, ngridmax), unew = (4, ncell) scratch. */
#include <stdint.h>
#include <stdlib.h>

static double dmax(double a, double b) { return a > b ? a : b; }

typedef struct {
  const int *son, *nbor;
  int64_t ncoarse, ngridmax;
} tree_t;

/* same-level neighbour of cell c (1-based) in direction dir (0:-x 1:+x 2:-y 3:+y 4:-z 5:+z); 0 if the
 * neighbouring oct does not exist */
static int64_t nbor_cell(const tree_t *T, int64_t c, int dir) {
  const int pos = (int)((c - T->ncoarse - 1) / T->ngridmax);
  const int64_t g = c - T->ncoarse - (int64_t)pos * T->ngridmax;
  const int axis = dir >> 1, up = dir & 1;
  const int bit = (pos >> axis) & 1;
  if (bit != up) return c + (up ? 1 : -1) * ((int64_t)(1 << axis) * T->ngridmax);
  const int nb = T->nbor[(int64_t)dir * T->ngridmax + g - 1];
  const int g2 = T->son[nb - 1];
  if (g2 == 0) return 0;
  return T->ncoarse + (int64_t)(pos ^ (1 << axis)) * T->ngridmax + g2;
}

void ora_rho_fine_hydro(int ilevel, int levelmin, int nvector, int ngrid_tot, const int *igrid, const double *xg,
                        const int *son, const int *nbor, const int *father, int64_t ngridmax, int64_t ncoarse,
                        double boxlen, double smallr, const double *dens, double *unew, double *rho,
                        double *multipole, double *rho_tot) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  tree_t T = {son, nbor, ncoarse, ngridmax};
  double dx = 1.0;
  for (int l = 0; l < ilevel; l++) dx *= 0.5;
  const double scale = boxlen;                 /* nx_loc = 1, skip_loc = 0 */
  const double dx_loc = dx * scale;
  const double vol_loc = dx_loc * dx_loc * dx_loc;
  double xc[8][3];
  for (int ind = 0; ind < 8; ind++) {
    const int iz = ind / 4, iy = (ind - 4 * iz) / 2, ix = ind - 2 * iy - 4 * iz;
    xc[ind][0] = ((double)ix - 0.5) * dx;
    xc[ind][1] = ((double)iy - 0.5) * dx;
    xc[ind][2] = ((double)iz - 0.5) * dx;
  }
  if (ilevel == levelmin) for (int d = 0; d < 4; d++) multipole[d] = 0.0;   /* rho_fine :28 */

  /* ---- multipole_fine ---- */
  for (int ind = 0; ind < 8; ind++)
    for (int i = 0; i < ngrid_tot; i++)
      for (int d = 0; d < 4; d++) unew[(int64_t)d * ncell + ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1] = 0.0;
  for (int i0 = 0; i0 < ngrid_tot; i0 += nvector) {
    const int ngrid = ngrid_tot - i0 < nvector ? ngrid_tot - i0 : nvector;
    for (int ind = 0; ind < 8; ind++) {
      for (int i = 0; i < ngrid; i++) {
        const int g = igrid[i0 + i];
        const int64_t c = ncoarse + (int64_t)ind * ngridmax + g - 1;
        if (son[c] != 0) continue;             /* split cells: not on a fully refined finest level */
        const double mm = dmax(dens[c], smallr) * vol_loc;
        unew[c] = unew[c] + mm;
      }
      for (int d = 0; d < 3; d++)
        for (int i = 0; i < ngrid; i++) {
          const int g = igrid[i0 + i];
          const int64_t c = ncoarse + (int64_t)ind * ngridmax + g - 1;
          if (son[c] != 0) continue;
          const double xx = (xg[(int64_t)d * ngridmax + g - 1] + xc[ind][d] - 0.0) * scale;
          const double mm = dmax(dens[c], smallr) * vol_loc;
          unew[(int64_t)(d + 1) * ncell + c] = unew[(int64_t)(d + 1) * ncell + c] + mm * xx;
        }
    }
  }

  /* ---- cic_from_multipole ---- */
  for (int ind = 0; ind < 8; ind++)
    for (int i = 0; i < ngrid_tot; i++) rho[ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1] = 0.0;
  int64_t *fc27 = (int64_t *)malloc(sizeof(int64_t) * 27 * (size_t)nvector);
  for (int i0 = 0; i0 < ngrid_tot; i0 += nvector) {
    const int np = ngrid_tot - i0 < nvector ? ngrid_tot - i0 : nvector;
    /* get3cubefather: the 3^3 father cells around the father cell of every oct, x fastest */
    for (int j = 0; j < np; j++) {
      const int64_t f0 = father[igrid[i0 + j] - 1];
      for (int t = 0; t < 27; t++) {
        const int d3[3] = {t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1};
        int64_t c = f0;
        for (int axis = 0; axis < 3 && c > 0; axis++)
          if (d3[axis] != 0) c = nbor_cell(&T, c, 2 * axis + (d3[axis] > 0 ? 1 : 0));
        fc27[(size_t)j * 27 + t] = c;
      }
    }
    for (int ind_son = 0; ind_son < 8; ind_son++) {
      const int64_t iskip_son = ncoarse + (int64_t)ind_son * ngridmax;
      if (ilevel == levelmin)
        for (int d = 0; d < 4; d++)
          for (int j = 0; j < np; j++) multipole[d] = multipole[d] + unew[(int64_t)d * ncell + iskip_son + igrid[i0 + j] - 1];
      /* the eight targets of every source cell are accumulated target by target, cell by cell */
      for (int ind = 0; ind < 8; ind++) {
        for (int j = 0; j < np; j++) {
          const int g = igrid[i0 + j];
          const int64_t cs = iskip_son + g - 1;
          double dd[3], dg[3];
          int ig[3], id[3];
          for (int d = 0; d < 3; d++) {
            double x = unew[(int64_t)(d + 1) * ncell + cs] / unew[cs];       /* centre of mass */
            x = x / scale + 0.0;
            x = x - (xg[(int64_t)d * ngridmax + g - 1] - 3.0 * dx);
            x = x / dx;
            dd[d] = x + 0.5;
            id[d] = (int)dd[d];
            dd[d] = dd[d] - id[d];
            dg[d] = 1.0 - dd[d];
            ig[d] = id[d] - 1;
          }
          const int bx = ind & 1, by = (ind >> 1) & 1, bz = (ind >> 2) & 1;
          const double vol = (bx ? dd[0] : dg[0]) * (by ? dd[1] : dg[1]) * (bz ? dd[2] : dg[2]);
          const int kx = bx ? id[0] : ig[0], ky = by ? id[1] : ig[1], kz = bz ? id[2] : ig[2];
          const int kg = (kx / 2) + 3 * (ky / 2) + 9 * (kz / 2);
          const int64_t fcell = fc27[(size_t)j * 27 + kg];
          const int gt = fcell > 0 ? son[fcell - 1] : 0;
          const int icell = (kx - 2 * (kx / 2)) + 2 * (ky - 2 * (ky / 2)) + 4 * (kz - 2 * (kz / 2));
          const double vol2 = unew[cs] * vol / vol_loc;
          if (gt > 0) {
            const int64_t ct = ncoarse + (int64_t)icell * ngridmax + gt - 1;
            rho[ct] = rho[ct] + vol2;
          }
        }
      }
    }
  }
  free(fc27);
  *rho_tot = multipole[0] / (scale * scale * scale);                           /* :179 */
}

/* ---------------------------------------------------------------------------------------------
 * rho_fine's hydro deposit on AMR levels: what one call rho_fine(ilevel,icount) with ilevel == levelmin or icount > 1
 * leaves in rho on every level it visits (pm/rho_fine.f90:45-60: i = nlevelmax .. ilevel: multipole_fine(i), then
 * cic_from_multipole(i)).
 *   multipole_fine(i) :666-820 -- leaf cells: m = max(rho,smallr)*vol, m*x at the cell centre; split cells: the sum of the
 *     eight children's multipoles, child by child (ind_son = 1..8) from zero;
 *   cic_from_multipole(i) / cic_cell :825-1142 -- rho of the level's octs reset, then every cell of the level (leaf or split)
 *     CIC-deposited at its centre of mass onto the level's own cells; a corner whose oct does not exist is dropped (ok(j),
 *     :1128-1139: that mass is counted on the coarser level through the father's multipole); multipole(1:4) += the cells of
 *     levelmin (:931-938).
 * lists: first[l - ilevel] .. first[l - ilevel + 1] of igrid_all = active(l)%igrid in list order, l = ilevel .. nlevelmax.
 * Periodic nx=ny=nz=1 box, 3-D, no particles, one rank.  unew = (4, ncell) scratch, rho = (ncell).
 * Pinned by tests/test_rho_fine_oracle.py on dumps of the unmodified reference in a self-gravitating AMR run
 * (oracle/dump_patch/rho_fine.f90 -> tests/golden/rho_fine_amr_ref.npz). */
void ora_rho_fine_amr(int ilevel, int nlevelmax, int levelmin, int nvector, const int *first, const int *igrid_all, const double *xg,
                      const int *son, const int *nbor, const int *father, int64_t ngridmax, int64_t ncoarse, double boxlen,
                      double smallr, const double *dens, double *unew, double *rho, double *multipole, double *rho_tot) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  tree_t T = {son, nbor, ncoarse, ngridmax};
  const double scale = boxlen;                 /* nx_loc = 1, skip_loc = 0 */
  if (ilevel == levelmin) for (int d = 0; d < 4; d++) multipole[d] = 0.0;   /* rho_fine :28 */
  int64_t *fc27 = (int64_t *)malloc(sizeof(int64_t) * 27 * (size_t)nvector);
  for (int lev = nlevelmax; lev >= ilevel; lev--) {
    const int *igrid = igrid_all + first[lev - ilevel];
    const int ngrid_tot = first[lev - ilevel + 1] - first[lev - ilevel];
    if (ngrid_tot == 0) continue;              /* (numbtot(1,lev) == 0: both routines return) */
    double dx = 1.0;
    for (int l = 0; l < lev; l++) dx *= 0.5;
    const double dx_loc = dx * scale;
    const double vol_loc = dx_loc * dx_loc * dx_loc;
    double xc[8][3];
    for (int ind = 0; ind < 8; ind++) {
      const int iz = ind / 4, iy = (ind - 4 * iz) / 2, ix = ind - 2 * iy - 4 * iz;
      xc[ind][0] = ((double)ix - 0.5) * dx;
      xc[ind][1] = ((double)iy - 0.5) * dx;
      xc[ind][2] = ((double)iz - 0.5) * dx;
    }
    /* ---- multipole_fine(lev) ---- */
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid_tot; i++)
        for (int d = 0; d < 4; d++) unew[(int64_t)d * ncell + ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1] = 0.0;
    for (int i0 = 0; i0 < ngrid_tot; i0 += nvector) {
      const int ngrid = ngrid_tot - i0 < nvector ? ngrid_tot - i0 : nvector;
      for (int ind = 0; ind < 8; ind++) {
        /* leaf cells (:737-757) */
        for (int d = 0; d < 4; d++)
          for (int i = 0; i < ngrid; i++) {
            const int g = igrid[i0 + i];
            const int64_t c = ncoarse + (int64_t)ind * ngridmax + g - 1;
            if (son[c] != 0) continue;
            const double mm = dmax(dens[c], smallr) * vol_loc;
            if (d == 0) {
              unew[c] = unew[c] + mm;
            } else {
              const double xx = (xg[(int64_t)(d - 1) * ngridmax + g - 1] + xc[ind][d - 1] - 0.0) * scale;
              unew[(int64_t)d * ncell + c] = unew[(int64_t)d * ncell + c] + mm * xx;
            }
          }
        /* split cells: children one after the other (:785-800) */
        for (int ind_son = 0; ind_son < 8; ind_son++)
          for (int d = 0; d < 4; d++)
            for (int i = 0; i < ngrid; i++) {
              const int g = igrid[i0 + i];
              const int64_t c = ncoarse + (int64_t)ind * ngridmax + g - 1;
              if (son[c] == 0) continue;
              const int64_t cs = ncoarse + (int64_t)ind_son * ngridmax + son[c] - 1;
              unew[(int64_t)d * ncell + c] = unew[(int64_t)d * ncell + c] + unew[(int64_t)d * ncell + cs];
            }
      }
    }
    /* ---- cic_from_multipole(lev) ---- */
    for (int ind = 0; ind < 8; ind++)
      for (int i = 0; i < ngrid_tot; i++) rho[ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1] = 0.0;
    for (int i0 = 0; i0 < ngrid_tot; i0 += nvector) {
      const int np = ngrid_tot - i0 < nvector ? ngrid_tot - i0 : nvector;
      for (int j = 0; j < np; j++) {
        const int64_t f0 = father[igrid[i0 + j] - 1];
        for (int t = 0; t < 27; t++) {
          const int d3[3] = {t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1};
          int64_t c = f0;
          for (int axis = 0; axis < 3 && c > 0; axis++)
            if (d3[axis] != 0) c = nbor_cell(&T, c, 2 * axis + (d3[axis] > 0 ? 1 : 0));
          fc27[(size_t)j * 27 + t] = c;
        }
      }
      for (int ind_son = 0; ind_son < 8; ind_son++) {
        const int64_t iskip_son = ncoarse + (int64_t)ind_son * ngridmax;
        if (lev == levelmin)
          for (int d = 0; d < 4; d++)
            for (int j = 0; j < np; j++) multipole[d] = multipole[d] + unew[(int64_t)d * ncell + iskip_son + igrid[i0 + j] - 1];
        for (int ind = 0; ind < 8; ind++) {
          for (int j = 0; j < np; j++) {
            const int g = igrid[i0 + j];
            const int64_t cs = iskip_son + g - 1;
            double dd[3], dg[3];
            int ig[3], id[3];
            for (int d = 0; d < 3; d++) {
              double x = unew[(int64_t)(d + 1) * ncell + cs] / unew[cs];       /* centre of mass */
              x = x / scale + 0.0;
              x = x - (xg[(int64_t)d * ngridmax + g - 1] - 3.0 * dx);
              x = x / dx;
              dd[d] = x + 0.5;
              id[d] = (int)dd[d];
              dd[d] = dd[d] - id[d];
              dg[d] = 1.0 - dd[d];
              ig[d] = id[d] - 1;
            }
            const int bx = ind & 1, by = (ind >> 1) & 1, bz = (ind >> 2) & 1;
            const double vol = (bx ? dd[0] : dg[0]) * (by ? dd[1] : dg[1]) * (bz ? dd[2] : dg[2]);
            const int kx = bx ? id[0] : ig[0], ky = by ? id[1] : ig[1], kz = bz ? id[2] : ig[2];
            const int kg = (kx / 2) + 3 * (ky / 2) + 9 * (kz / 2);
            const int64_t fcell = fc27[(size_t)j * 27 + kg];
            const int gt = fcell > 0 ? son[fcell - 1] : 0;
            const int icell = (kx - 2 * (kx / 2)) + 2 * (ky - 2 * (ky / 2)) + 4 * (kz - 2 * (kz / 2));
            const double vol2 = unew[cs] * vol / vol_loc;
            if (gt > 0) {
              const int64_t ct = ncoarse + (int64_t)icell * ngridmax + gt - 1;
              rho[ct] = rho[ct] + vol2;
            }
          }
        }
      }
    }
  }
  free(fc27);
  *rho_tot = multipole[0] / (scale * scale * scale);                           /* :179 */
}

/* ---------------------------------------------------------------------------------------------
 * The same deposit as a GATHER (the shape a device kernel needs: one thread per target cell, no
 * atomics): every contribution is tagged with the position it has in the reference's loop nest
 * (batch of nvector octs, ind_son, CIC corner, oct in batch); a target sorts what it receives by
 * that tag and adds it in order.  The sources are visited in REVERSE here on purpose: the result
 * must not depend on the order in which the contributions are produced.  tests/test_rho_fine_oracle.py
 * checks it against ora_rho_fine_hydro and the reference's dumps, bit for bit.
 * unew must hold the multipoles (call ora_rho_fine_hydro first); rho_out receives the deposit. */
typedef struct { int64_t key; double val; } contrib_t;

static int cmp_contrib(const void *a, const void *b) {
  const int64_t ka = ((const contrib_t *)a)->key, kb = ((const contrib_t *)b)->key;
  return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

void ora_rho_deposit_gather(int ilevel, int nvector, int ngrid_tot, const int *igrid, const double *xg, const int *son,
                            const int *nbor, const int *father, int64_t ngridmax, int64_t ncoarse, double boxlen,
                            const double *unew, double *rho_out) {
  const int64_t ncell = ncoarse + 8 * ngridmax;
  tree_t T = {son, nbor, ncoarse, ngridmax};
  double dx = 1.0;
  for (int l = 0; l < ilevel; l++) dx *= 0.5;
  const double scale = boxlen, dx_loc = dx * scale, vol_loc = dx_loc * dx_loc * dx_loc;
  enum { CAP = 64 };
  contrib_t *box = (contrib_t *)malloc(sizeof(contrib_t) * CAP * (size_t)ncell);
  int *cnt = (int *)calloc((size_t)ncell, sizeof(int));
  for (int i = ngrid_tot - 1; i >= 0; i--) {                 /* sources in reverse list order */
    const int g = igrid[i];
    const int64_t batch = i / nvector, j = i % nvector;
    int64_t fc27[27];
    const int64_t f0 = father[g - 1];
    for (int t = 0; t < 27; t++) {
      const int d3[3] = {t % 3 - 1, (t / 3) % 3 - 1, t / 9 - 1};
      int64_t c = f0;
      for (int axis = 0; axis < 3 && c > 0; axis++)
        if (d3[axis] != 0) c = nbor_cell(&T, c, 2 * axis + (d3[axis] > 0 ? 1 : 0));
      fc27[t] = c;
    }
    for (int ind_son = 7; ind_son >= 0; ind_son--) {
      const int64_t cs = ncoarse + (int64_t)ind_son * ngridmax + g - 1;
      double dd[3], dg[3];
      int ig[3], id[3];
      for (int d = 0; d < 3; d++) {
        double x = unew[(int64_t)(d + 1) * ncell + cs] / unew[cs];
        x = x / scale + 0.0;
        x = x - (xg[(int64_t)d * ngridmax + g - 1] - 3.0 * dx);
        x = x / dx;
        dd[d] = x + 0.5;
        id[d] = (int)dd[d];
        dd[d] = dd[d] - id[d];
        dg[d] = 1.0 - dd[d];
        ig[d] = id[d] - 1;
      }
      for (int ind = 7; ind >= 0; ind--) {
        const int bx = ind & 1, by = (ind >> 1) & 1, bz = (ind >> 2) & 1;
        const double vol = (bx ? dd[0] : dg[0]) * (by ? dd[1] : dg[1]) * (bz ? dd[2] : dg[2]);
        const int kx = bx ? id[0] : ig[0], ky = by ? id[1] : ig[1], kz = bz ? id[2] : ig[2];
        const int64_t fcell = fc27[(kx / 2) + 3 * (ky / 2) + 9 * (kz / 2)];
        const int gt = fcell > 0 ? son[fcell - 1] : 0;
        if (gt <= 0) continue;
        const int icell = (kx - 2 * (kx / 2)) + 2 * (ky - 2 * (ky / 2)) + 4 * (kz - 2 * (kz / 2));
        const int64_t ct = ncoarse + (int64_t)icell * ngridmax + gt - 1;
        contrib_t *b = box + (size_t)ct * CAP + cnt[ct]++;
        b->key = ((batch * 8 + ind_son) * 8 + ind) * nvector + j;
        b->val = unew[cs] * vol / vol_loc;
      }
    }
  }
  for (int ind = 0; ind < 8; ind++)
    for (int i = 0; i < ngrid_tot; i++) {
      const int64_t ct = ncoarse + (int64_t)ind * ngridmax + igrid[i] - 1;
      qsort(box + (size_t)ct * CAP, (size_t)cnt[ct], sizeof(contrib_t), cmp_contrib);
      double r = 0.0;
      for (int k = 0; k < cnt[ct]; k++) r = r + box[(size_t)ct * CAP + k].val;
      rho_out[ct] = r;
    }
  free(box);
  free(cnt);
}
