"""Namelist of the MHD end-to-end tests: a blast in a magnetised medium on a uniform periodic level (the 'square'
regions of mhd/init_flow_fine.f90 with A_region / B_region / C_region), SOLVER=mhd, NVAR=8."""

MHD_NML = """&RUN_PARAMS
hydro=.true.
nrestart=0
ncontrol=1
nremap=0
nsubcycle=10*1
nstepmax={nstep}
/

&AMR_PARAMS
levelmin={level}
levelmax={level}
ngridtot={ngridtot}
nexpand=1
boxlen=1.0
/

&INIT_PARAMS
nregion=3
region_type(1)='square'
region_type(2)='square'
region_type(3)='square'
x_center=0.5,0.5,0.3
y_center=0.5,0.5,0.6
z_center=0.5,0.5,0.4
length_x=10.0,0.25,0.2
length_y=10.0,0.25,0.3
length_z=10.0,0.25,0.2
exp_region=10.0,2.0,10.0
d_region=1.0,1.0,3.0
u_region=0.1,0.1,-0.3
v_region=-0.2,-0.2,0.2
w_region=0.05,0.05,0.1
p_region=0.1,10.0,0.1
A_region=1.0,1.0,1.0
B_region=0.5,0.5,0.5
C_region=-0.3,-0.3,-0.3
/

&OUTPUT_PARAMS
foutput={nstep}
noutput=1
tout=10.0
/

&HYDRO_PARAMS
gamma=1.6666667
courant_factor=0.8
slope_type={slope_type}
riemann='{riemann}'
riemann2d='{riemann2d}'
/
"""


def mhd_namelist(level=5, nstep=8, riemann="llf", riemann2d="llf", slope_type=1):
    ngridtot = int(1.3 * sum(8 ** l for l in range(level))) + 1000
    return MHD_NML.format(level=level, nstep=nstep, ngridtot=ngridtot, riemann=riemann, riemann2d=riemann2d, slope_type=slope_type)
