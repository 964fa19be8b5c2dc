!> Fortran host side of the MI355X-native Radtran hot path.
!>
!> Drop-in shaped like the reference's `module clima_radtran` (src/radtran/clima_radtran.f90):
!> `type(Radtran)` with the same public fields (:51-72), the same type-bound procedures
!> `radiate` (:221-318), `TOA_fluxes` (:320-342), `apply_radiation_enhancement` (:402-411),
!> `set_bolometric_flux` / `bolometric_flux` / `skin_temperature` /
!> `equilibrium_temperature` (:345-382), and the result holders `wrk_ir`, `wrk_sol`
!> (`type ClimaRadtranWrk`, :11-25), `ir`, `sol` (RTChannel), `f_total`.  Errors use the
!> reference convention: `character(:), allocatable, intent(out) :: err`, allocated <=> failure.
!>
!> Everything numerical happens in hand-written HIP kernels behind the C ABI of
!> include/clima_radtran_hip.h; this module only marshals arguments (ISO_C_BINDING).
!> Construction takes the loaded tables instead of file names because the HDF5/YAML loaders
!> (clima_radtran_types_create.f90) stay on the host side of the boundary:
!>     call rad%begin(nz, species_count, particle_count, wavl, err)
!>     call rad%add_ktable(...) ; call rad%add_xsection(...) ; ...
!>     call rad%finish(num_zenith_angles, surface_albedo, err)
module clima_radtran_hip
  use iso_c_binding
  implicit none
  private

  public :: Radtran, ClimaRadtranWrk, RTChannel, dp
  public :: radtran_set_device, radtran_comm_unique_id, comm_id_bytes
  public :: CIAXsection, RayleighXsection, AbsorptionXsection, PhotolysisXsection

  integer, parameter :: dp = c_double
  integer, parameter :: err_len = 1024
  integer, parameter :: comm_id_bytes = 128   !! CLIMA_COMM_ID_BYTES of include/clima_radtran_hip.h
  ! enum of src/radtran/clima_radtran_types.f90:40-42
  integer, parameter :: CIAXsection = 0, RayleighXsection = 1, AbsorptionXsection = 2, PhotolysisXsection = 3

  type :: ClimaRadtranWrk
    real(dp), allocatable :: fup_a(:,:), fdn_a(:,:) !! (nz+1,nw) mW/m2/Hz
    real(dp), allocatable :: fup_n(:), fdn_n(:)     !! (nz+1) mW/m2
    real(dp), allocatable :: amean(:,:)             !! (nz+1,nw) photons/cm^2/s (solar)
    real(dp), allocatable :: tau_band(:,:)          !! (nz,nw)
  end type

  type :: RTChannel
    integer :: nw = 0
    real(dp), allocatable :: wavl(:), freq(:)
  end type

  type :: Radtran
    integer :: ng = 0   !! number of gases
    integer :: np = 0   !! number of particles
    integer :: nz = 0
    type(RTChannel) :: ir, sol
    real(dp) :: diurnal_fac = 0.5_dp
    real(dp), allocatable :: zenith_u(:), zenith_weights(:)
    real(dp), allocatable :: surface_albedo(:), surface_emissivity(:)
    logical :: has_hard_surface = .true.
    real(dp) :: ir_tau_min = 1.0e-6_dp
    real(dp), allocatable :: photons_sol(:)
    real(dp) :: photon_scale_factor = 1.0_dp
    type(ClimaRadtranWrk) :: wrk_ir, wrk_sol
    real(dp), allocatable :: f_total(:)
    !> copy the per-bin spectra (fup_a, fdn_a, amean, tau_band) back after every radiate;
    !> set to .false. when only the level fluxes are needed (saves ~7 MB of PCIe per call)
    logical :: sync_spectra = .true.
    !> radiate_ir_batch: page-lock the three result arrays when the same ones come again (the Jacobian's work arrays) and
    !> let the device fill them directly (include/clima_radtran_hip.h, radtran_batch_pin_results_set).  The arrays must then
    !> stay allocated until `destroy`.  Off by default.
    logical :: pin_batch_results = .false.
    type(c_ptr) :: handle = c_null_ptr
  contains
    procedure :: begin => Radtran_begin
    procedure :: add_ktable => Radtran_add_ktable
    procedure :: add_xsection => Radtran_add_xsection
    procedure :: set_water_continuum => Radtran_set_water_continuum
    procedure :: add_particle => Radtran_add_particle
    procedure :: set_channels => Radtran_set_channels
    procedure :: set_photons_sol => Radtran_set_photons_sol
    procedure :: finish => Radtran_finish
    procedure :: radiate => Radtran_radiate
    procedure :: TOA_fluxes => Radtran_TOA_fluxes
    procedure :: set_bolometric_flux => Radtran_set_bolometric_flux
    procedure :: bolometric_flux => Radtran_bolometric_flux
    procedure :: skin_temperature => Radtran_skin_temperature
    procedure :: equilibrium_temperature => Radtran_equilibrium_temperature
    procedure :: apply_radiation_enhancement => Radtran_apply_radiation_enhancement
    procedure :: radiate_ir_batch => Radtran_radiate_ir_batch
    procedure :: TOA_fluxes_batch => Radtran_TOA_fluxes_batch
    procedure :: opacities2yaml => Radtran_opacities2yaml
    procedure :: set_names => Radtran_set_names
    procedure :: set_custom_optical_properties => Radtran_set_custom_optical_properties
    procedure :: unset_custom_optical_properties => Radtran_unset_custom_optical_properties
    procedure :: destroy => Radtran_destroy
    !> The library's own multi-GPU step (one process per GPU, include/clima_radtran_hip.h radtran_comm_*):
    !> after `comm_init` / `comm_init_file`, `radiate` and `TOA_fluxes` work on this rank's spectral bins
    !> and end with one RCCL all-reduce of the level fluxes; `wrk_ir%fup_n` ... `f_total` are the whole
    !> spectrum's on every rank (the sum over bins of src/radtran/clima_radtran_radiate.f90:184-192).
    procedure :: comm_init => Radtran_comm_init
    procedure :: comm_init_file => Radtran_comm_init_file
    procedure :: comm_destroy => Radtran_comm_destroy
    procedure :: set_ir_green => Radtran_set_ir_green
    procedure :: release_pinned => Radtran_release_pinned
    procedure :: ir_green_batches => Radtran_ir_green_batches
  end type

  interface
    subroutine c_allocate_radtran(ptr) bind(c, name="allocate_radtran")
      import; type(c_ptr), intent(out) :: ptr
    end subroutine
    subroutine c_deallocate_radtran(ptr) bind(c, name="deallocate_radtran")
      import; type(c_ptr), value :: ptr
    end subroutine
    subroutine c_radtran_create_begin(ptr, nz, nsp, np, nw, wavl, err) bind(c, name="radtran_create_begin")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: nz, nsp, np, nw
      real(c_double), intent(in) :: wavl(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_add_ktable(ptr, sp_ind, ngauss, weights, npress, log10P, ntemp, temp, log10k, err) bind(c, name="radtran_add_ktable")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: sp_ind, ngauss, npress, ntemp
      real(c_double), intent(in) :: weights(*), log10P(*), temp(*), log10k(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_add_xsection(ptr, xs_type, dim, sp_ind1, sp_ind2, ntemp, temp, data, err) bind(c, name="radtran_add_xsection")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: xs_type, dim, sp_ind1, sp_ind2, ntemp
      real(c_double), intent(in) :: temp(*), data(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_set_water_continuum(ptr, LH2O, ntemp, temp, h2o, foreign, err) bind(c, name="radtran_set_water_continuum")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: LH2O, ntemp
      real(c_double), intent(in) :: temp(*), h2o(*), foreign(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_add_particle(ptr, p_ind, nrad, radii, w0, qext, gt, err) bind(c, name="radtran_add_particle")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: p_ind, nrad
      real(c_double), intent(in) :: radii(*), w0(*), qext(*), gt(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_set_channels(ptr, n_ir, ir_wavl, n_sol, sol_wavl, err) bind(c, name="radtran_set_channels")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: n_ir, n_sol
      real(c_double), intent(in) :: ir_wavl(*), sol_wavl(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_set_photons_sol(ptr, n, photons_sol, err) bind(c, name="radtran_set_photons_sol")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: n
      real(c_double), intent(in) :: photons_sol(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_create_end(ptr, nzen, albedo, err) bind(c, name="radtran_create_end")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: nzen
      real(c_double), intent(in) :: albedo
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_radiate_wrapper(ptr, T_surface, dim_T, T, dim_P, P, dim1_d, dim2_d, densities, &
                                       dim_dz, dz, has_particles, dim1_p, dim2_p, pdensities, dim1_r, dim2_r, radii, &
                                       compute_solar, compute_opacity, err) bind(c, name="radtran_radiate_wrapper")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: T_surface
      integer(c_int), intent(in) :: dim_T, dim_P, dim1_d, dim2_d, dim_dz, has_particles, dim1_p, dim2_p, dim1_r, dim2_r
      real(c_double), intent(in) :: T(*), P(*), densities(*), dz(*), pdensities(*), radii(*)
      integer(c_int), intent(in) :: compute_solar, compute_opacity
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_apply_radiation_enhancement(ptr, rad_enhancement) bind(c, name="radtran_apply_radiation_enhancement")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: rad_enhancement
    end subroutine
    subroutine c_radtran_toa_fluxes_batch(ptr, ncol, T_surface, T, P, densities, dz, has_particles, pdensities, radii, &
                                          ISR, OLR, fluxes, err) bind(c, name="radtran_toa_fluxes_batch")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: ncol, has_particles
      real(c_double), intent(in) :: T_surface(*), T(*), P(*), densities(*), dz(*), pdensities(*), radii(*)
      real(c_double), intent(out) :: ISR(*), OLR(*), fluxes(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_radiate_ir_batch(ptr, ncol, T_surface, dim1_T, dim2_T, T, fup_n, fdn_n, f_total, err) &
                                          bind(c, name="radtran_radiate_ir_batch")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: ncol, dim1_T, dim2_T
      real(c_double), intent(in) :: T_surface(*), T(*)
      real(c_double), intent(out) :: fup_n(*), fdn_n(*), f_total(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_opacities2yaml_wrapper_1(ptr, out_len, out_cp) bind(c, name="radtran_opacities2yaml_wrapper_1")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(out) :: out_len
      type(c_ptr), intent(out) :: out_cp
    end subroutine
    subroutine c_radtran_opacities2yaml_wrapper_2(ptr, out_cp, out_len, out_c) bind(c, name="radtran_opacities2yaml_wrapper_2")
      import; type(c_ptr), value :: ptr
      type(c_ptr), intent(inout) :: out_cp
      integer(c_int), intent(in) :: out_len
      character(c_char), intent(out) :: out_c(*)
    end subroutine
    subroutine c_radtran_set_names(ptr, species_names, particle_names, err) bind(c, name="radtran_set_names")
      import; type(c_ptr), value :: ptr
      character(c_char), intent(in) :: species_names(*), particle_names(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_set_custom_optical_properties(ptr, dim_wv, wv, dim_P, P, dim1_t, dim2_t, dtau_dz, &
                                                       dim1_w, dim2_w, w0, dim1_g, dim2_g, g0, err) &
                                                       bind(c, name="radtran_set_custom_optical_properties")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim_wv, dim_P, dim1_t, dim2_t, dim1_w, dim2_w, dim1_g, dim2_g
      real(c_double), intent(in) :: wv(*), P(*), dtau_dz(*), w0(*), g0(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_unset_custom_optical_properties(ptr) bind(c, name="radtran_unset_custom_optical_properties")
      import; type(c_ptr), value :: ptr
    end subroutine
    subroutine c_radtran_set_bolometric_flux_wrapper(ptr, flux) bind(c, name="radtran_set_bolometric_flux_wrapper")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: flux
    end subroutine
    subroutine c_radtran_bolometric_flux_wrapper(ptr, flux) bind(c, name="radtran_bolometric_flux_wrapper")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(out) :: flux
    end subroutine
    subroutine c_radtran_skin_temperature_wrapper(ptr, bond_albedo, T_skin) bind(c, name="radtran_skin_temperature_wrapper")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: bond_albedo
      real(c_double), intent(out) :: T_skin
    end subroutine
    subroutine c_radtran_equilibrium_temperature_wrapper(ptr, bond_albedo, T_eq) bind(c, name="radtran_equilibrium_temperature_wrapper")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: bond_albedo
      real(c_double), intent(out) :: T_eq
    end subroutine
    subroutine c_radtran_zenith_u_get(ptr, dim1, arr) bind(c, name="radtran_zenith_u_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_radtran_zenith_u_set(ptr, dim1, arr) bind(c, name="radtran_zenith_u_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(in) :: arr(*)
    end subroutine
    subroutine c_radtran_zenith_weights_get(ptr, dim1, arr) bind(c, name="radtran_zenith_weights_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_radtran_zenith_weights_set(ptr, dim1, arr) bind(c, name="radtran_zenith_weights_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(in) :: arr(*)
    end subroutine
    subroutine c_radtran_surface_albedo_set(ptr, dim1, arr) bind(c, name="radtran_surface_albedo_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(in) :: arr(*)
    end subroutine
    subroutine c_radtran_surface_emissivity_set(ptr, dim1, arr) bind(c, name="radtran_surface_emissivity_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(in) :: arr(*)
    end subroutine
    subroutine c_radtran_has_hard_surface_set(ptr, val) bind(c, name="radtran_has_hard_surface_set")
      import; type(c_ptr), value :: ptr
      logical(c_bool), intent(in) :: val  ! as clima/fortran/Radtran.f90:220-227
    end subroutine
    subroutine c_radtran_photon_scale_factor_set(ptr, val) bind(c, name="radtran_photon_scale_factor_set")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: val
    end subroutine
    subroutine c_radtran_photon_scale_factor_get(ptr, val) bind(c, name="radtran_photon_scale_factor_get")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(out) :: val
    end subroutine
    subroutine c_radtran_ir_tau_min_set(ptr, val) bind(c, name="radtran_ir_tau_min_set")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: val
    end subroutine
    subroutine c_radtran_diurnal_fac_set(ptr, val) bind(c, name="radtran_diurnal_fac_set")
      import; type(c_ptr), value :: ptr
      real(c_double), intent(in) :: val
    end subroutine
    subroutine c_radtran_ir_get(ptr, ptr1) bind(c, name="radtran_ir_get")
      import; type(c_ptr), value :: ptr
      type(c_ptr), intent(out) :: ptr1
    end subroutine
    subroutine c_radtran_sol_get(ptr, ptr1) bind(c, name="radtran_sol_get")
      import; type(c_ptr), value :: ptr
      type(c_ptr), intent(out) :: ptr1
    end subroutine
    subroutine c_radtran_wrk_ir_get(ptr, ptr1) bind(c, name="radtran_wrk_ir_get")
      import; type(c_ptr), value :: ptr
      type(c_ptr), intent(out) :: ptr1
    end subroutine
    subroutine c_radtran_wrk_sol_get(ptr, ptr1) bind(c, name="radtran_wrk_sol_get")
      import; type(c_ptr), value :: ptr
      type(c_ptr), intent(out) :: ptr1
    end subroutine
    subroutine c_radtran_f_total_get(ptr, dim1, arr) bind(c, name="radtran_f_total_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_rtchannel_wavl_get_size(ptr, dim1) bind(c, name="rtchannel_wavl_get_size")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(out) :: dim1
    end subroutine
    subroutine c_rtchannel_wavl_get(ptr, dim1, arr) bind(c, name="rtchannel_wavl_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_rtchannel_freq_get(ptr, dim1, arr) bind(c, name="rtchannel_freq_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_climaradtranwrk_fup_a_get(ptr, dim1, dim2, arr) bind(c, name="climaradtranwrk_fup_a_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1, dim2
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_climaradtranwrk_fdn_a_get(ptr, dim1, dim2, arr) bind(c, name="climaradtranwrk_fdn_a_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1, dim2
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_climaradtranwrk_amean_get(ptr, dim1, dim2, arr) bind(c, name="climaradtranwrk_amean_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1, dim2
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_climaradtranwrk_tau_band_get(ptr, dim1, dim2, arr) bind(c, name="climaradtranwrk_tau_band_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1, dim2
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_radtran_spectra_get_all(ptr, do_solar, nlev, nw_ir, nw_sol, ir_fup_a, ir_fdn_a, ir_tau_band, &
                                         sol_fup_a, sol_fdn_a, sol_amean, sol_tau_band, err) bind(c, name="radtran_spectra_get_all")
      import; type(c_ptr), value :: ptr
      logical(c_bool), intent(in) :: do_solar
      integer(c_int), intent(in) :: nlev, nw_ir, nw_sol
      real(c_double), intent(out) :: ir_fup_a(*), ir_fdn_a(*), ir_tau_band(*)
      real(c_double), intent(out) :: sol_fup_a(*), sol_fdn_a(*), sol_amean(*), sol_tau_band(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_create_from_files(ptr, settings_file, star_file, num_zenith_angles, surface_albedo, nz, datadir, err) &
        bind(c, name="radtran_create_from_files")
      import; type(c_ptr), value :: ptr
      character(c_char), intent(in) :: settings_file(*), star_file(*), datadir(*)
      integer(c_int), intent(in) :: num_zenith_angles, nz
      real(c_double), intent(in) :: surface_albedo
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_dims_get(ptr, nz, nsp, np, nw, ngauss) bind(c, name="radtran_dims_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(out) :: nz, nsp, np, nw, ngauss
    end subroutine
    subroutine c_radtran_photons_sol_get(ptr, dim1, arr) bind(c, name="radtran_photons_sol_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_radtran_spectra_release(ptr) bind(c, name="radtran_spectra_release")
      import; type(c_ptr), value :: ptr
    end subroutine
    subroutine c_climaradtranwrk_fup_n_get(ptr, dim1, arr) bind(c, name="climaradtranwrk_fup_n_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
    subroutine c_radtran_set_device(device, err) bind(c, name="radtran_set_device")
      import; integer(c_int), intent(in) :: device
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_comm_unique_id(id, err) bind(c, name="radtran_comm_unique_id")
      import; character(c_char), intent(out) :: id(*), err(*)
    end subroutine
    subroutine c_radtran_comm_init_rank(ptr, nranks, rank, id, err) bind(c, name="radtran_comm_init_rank")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: nranks, rank
      character(c_char), intent(in) :: id(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_comm_init_file(ptr, nranks, rank, path, err) bind(c, name="radtran_comm_init_file")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: nranks, rank
      character(c_char), intent(in) :: path(*)
      character(c_char), intent(out) :: err(*)
    end subroutine
    subroutine c_radtran_comm_destroy(ptr) bind(c, name="radtran_comm_destroy")
      import; type(c_ptr), value :: ptr
    end subroutine
    subroutine c_radtran_batch_pin_results_set(ptr, flag) bind(c, name="radtran_batch_pin_results_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: flag
    end subroutine
    subroutine c_radtran_ir_green_set(ptr, mode) bind(c, name="radtran_ir_green_set")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: mode
    end subroutine
    subroutine c_radtran_ir_green_get(ptr, mode, batches) bind(c, name="radtran_ir_green_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(out) :: mode, batches
    end subroutine
    subroutine c_climaradtranwrk_fdn_n_get(ptr, dim1, arr) bind(c, name="climaradtranwrk_fdn_n_get")
      import; type(c_ptr), value :: ptr
      integer(c_int), intent(in) :: dim1
      real(c_double), intent(out) :: arr(*)
    end subroutine
  end interface

  !> `rad = Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`: the reference's constructor
  !> (src/radtran/clima_radtran.f90:98-126) with the files read behind the C ABI (radtran_create_from_files: the
  !> settings YAML's optical-properties block, the stellar spectrum, a photochem_clima_data-style directory), for hosts
  !> that do not link the reference's own loaders (src/radtran/clima_radtran_types_create.f90).
  interface Radtran
    module procedure :: create_Radtran_from_files
  end interface

contains

  !> reference convention: allocated err <=> failure (clima/fortran/AdiabatClimate.f90:185-188 reversed)
  subroutine take_err(err_c, err)
    character(c_char), intent(in) :: err_c(err_len+1)
    character(:), allocatable, intent(out) :: err
    integer :: i, n
    n = 0
    do i = 1, err_len
      if (err_c(i) == c_null_char) exit
      n = n + 1
    enddo
    if (n > 0) then
      allocate(character(n) :: err)
      do i = 1, n
        err(i:i) = err_c(i)
      enddo
    endif
  end subroutine

  subroutine Radtran_begin(self, nz, nsp, np, wavl, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: nz, nsp, np
    real(dp), intent(in) :: wavl(:) !! (nw+1) nm
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    if (c_associated(self%handle)) call c_deallocate_radtran(self%handle)
    call c_allocate_radtran(self%handle)
    self%nz = nz; self%ng = nsp; self%np = np
    call c_radtran_create_begin(self%handle, nz, nsp, np, size(wavl)-1, wavl, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_add_ktable(self, sp_ind, weights, log10P, temp, log10k, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: sp_ind !! 1-based species index (Ktable%sp_ind)
    real(dp), intent(in) :: weights(:), log10P(:), temp(:)
    real(dp), intent(in) :: log10k(:,:,:,:) !! (ngauss,npress,ntemp,nwav), types_create.f90:1349-1358
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_add_ktable(self%handle, sp_ind, size(weights), weights, size(log10P), log10P, &
                            size(temp), temp, log10k, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_add_xsection(self, xs_type, sp_ind, xs_0d, temp, log10_xs_1d, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: xs_type
    integer, intent(in) :: sp_ind(:) !! 1 or 2 species (Xsection%sp_ind)
    real(dp), optional, intent(in) :: xs_0d(:)          !! (nw)
    real(dp), optional, intent(in) :: temp(:)           !! (ntemp)
    real(dp), optional, intent(in) :: log10_xs_1d(:,:)  !! (ntemp,nw)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    integer :: sp2
    real(dp) :: dummy(1)
    sp2 = 0
    if (size(sp_ind) > 1) sp2 = sp_ind(2)
    dummy = 0.0_dp
    if (present(xs_0d)) then
      call c_radtran_add_xsection(self%handle, xs_type, 0, sp_ind(1), sp2, 0, dummy, xs_0d, err_c)
    elseif (present(temp) .and. present(log10_xs_1d)) then
      call c_radtran_add_xsection(self%handle, xs_type, 1, sp_ind(1), sp2, size(temp), temp, log10_xs_1d, err_c)
    else
      err = 'add_xsection needs either xs_0d or temp and log10_xs_1d'
      return
    endif
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_set_water_continuum(self, LH2O, temp, log10_xs_H2O, log10_xs_foreign, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: LH2O
    real(dp), intent(in) :: temp(:), log10_xs_H2O(:,:), log10_xs_foreign(:,:) !! (ntemp,nw)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_set_water_continuum(self%handle, LH2O, size(temp), temp, log10_xs_H2O, log10_xs_foreign, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_add_particle(self, p_ind, radii, w0, qext, gt, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: p_ind
    real(dp), intent(in) :: radii(:), w0(:,:), qext(:,:), gt(:,:) !! (nrad,nw)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_add_particle(self%handle, p_ind, size(radii), radii, w0, qext, gt, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_set_channels(self, ir_wavl, sol_wavl, err)
    class(Radtran), intent(inout) :: self
    real(dp), intent(in) :: ir_wavl(:), sol_wavl(:)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_set_channels(self%handle, size(ir_wavl), ir_wavl, size(sol_wavl), sol_wavl, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_set_photons_sol(self, photons_sol, err)
    class(Radtran), intent(inout) :: self
    real(dp), intent(in) :: photons_sol(:)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_set_photons_sol(self%handle, size(photons_sol), photons_sol, err_c)
    call take_err(err_c, err)
    if (.not. allocated(err)) self%photons_sol = photons_sol
  end subroutine

  !> rest of create_Radtran_2 (clima_radtran.f90:162-217)
  subroutine Radtran_finish(self, num_zenith_angles, surface_albedo, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: num_zenith_angles
    real(dp), intent(in) :: surface_albedo
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_create_end(self%handle, num_zenith_angles, surface_albedo, err_c)
    call take_err(err_c, err)
    if (allocated(err)) return
    call adopt_handle(self, num_zenith_angles, surface_albedo)
  end subroutine

  !> the module's side of a constructed handle: public fields and result arrays (clima_radtran.f90:160-214)
  subroutine adopt_handle(self, num_zenith_angles, surface_albedo)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: num_zenith_angles
    real(dp), intent(in) :: surface_albedo
    type(c_ptr) :: p
    integer(c_int) :: n
    integer :: nz
    nz = self%nz
    allocate(self%zenith_u(num_zenith_angles), self%zenith_weights(num_zenith_angles))
    call c_radtran_zenith_u_get(self%handle, num_zenith_angles, self%zenith_u)
    call c_radtran_zenith_weights_get(self%handle, num_zenith_angles, self%zenith_weights)
    call c_radtran_ir_get(self%handle, p)
    call c_rtchannel_wavl_get_size(p, n)
    self%ir%nw = n - 1
    allocate(self%ir%wavl(n), self%ir%freq(n))
    call c_rtchannel_wavl_get(p, n, self%ir%wavl)
    call c_rtchannel_freq_get(p, n, self%ir%freq)
    call c_radtran_sol_get(self%handle, p)
    call c_rtchannel_wavl_get_size(p, n)
    self%sol%nw = n - 1
    allocate(self%sol%wavl(n), self%sol%freq(n))
    call c_rtchannel_wavl_get(p, n, self%sol%wavl)
    call c_rtchannel_freq_get(p, n, self%sol%freq)
    allocate(self%surface_albedo(self%sol%nw)); self%surface_albedo = surface_albedo
    allocate(self%surface_emissivity(self%ir%nw)); self%surface_emissivity = 1.0_dp
    if (.not. allocated(self%photons_sol)) then
      allocate(self%photons_sol(self%sol%nw)); self%photons_sol = 0.0_dp
    endif
    call alloc_wrk(self%wrk_ir, nz, self%ir%nw)
    call alloc_wrk(self%wrk_sol, nz, self%sol%nw)
    allocate(self%f_total(nz+1)); self%f_total = 0.0_dp
  contains
    subroutine alloc_wrk(w, nz, nw)
      type(ClimaRadtranWrk), intent(inout) :: w
      integer, intent(in) :: nz, nw
      allocate(w%fup_a(nz+1,nw), w%fdn_a(nz+1,nw), w%amean(nz+1,nw), w%tau_band(nz,nw), w%fup_n(nz+1), w%fdn_n(nz+1))
      w%fup_a = 0.0_dp; w%fdn_a = 0.0_dp; w%amean = 0.0_dp; w%tau_band = 0.0_dp; w%fup_n = 0.0_dp; w%fdn_n = 0.0_dp
    end subroutine
  end subroutine

  function create_Radtran_from_files(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err) result(rad)
    character(*), intent(in) :: settings_f, star_f, datadir
    integer, intent(in) :: num_zenith_angles, nz
    real(dp), intent(in) :: surface_albedo
    character(:), allocatable, intent(out) :: err
    type(Radtran) :: rad
    character(c_char) :: err_c(err_len+1)
    integer(c_int) :: nz_c, nsp, np, nw, ngauss
    call c_allocate_radtran(rad%handle)
    call c_radtran_create_from_files(rad%handle, cstr(settings_f), cstr(star_f), int(num_zenith_angles, c_int), surface_albedo, &
                                     int(nz, c_int), cstr(datadir), err_c)
    call take_err(err_c, err)
    if (allocated(err)) then
      call c_deallocate_radtran(rad%handle)
      rad%handle = c_null_ptr
      return
    endif
    call c_radtran_dims_get(rad%handle, nz_c, nsp, np, nw, ngauss)
    rad%nz = nz_c; rad%ng = nsp; rad%np = np
    call adopt_handle(rad, num_zenith_angles, surface_albedo)
    call c_radtran_photons_sol_get(rad%handle, size(rad%photons_sol), rad%photons_sol)   ! what the star file gave
  contains
    function cstr(s) result(c)
      character(*), intent(in) :: s
      character(c_char) :: c(len_trim(s)+1)
      integer :: i
      do i = 1, len_trim(s)
        c(i) = s(i:i)
      enddo
      c(len_trim(s)+1) = c_null_char
    end function
  end function

  !> the public fields are read on every radiate (clima_radtran.f90:262-313)
  subroutine push_fields(self)
    class(Radtran), intent(inout) :: self
    logical(c_bool) :: hs
    call c_radtran_zenith_u_set(self%handle, size(self%zenith_u), self%zenith_u)
    call c_radtran_zenith_weights_set(self%handle, size(self%zenith_weights), self%zenith_weights)
    call c_radtran_surface_albedo_set(self%handle, size(self%surface_albedo), self%surface_albedo)
    call c_radtran_surface_emissivity_set(self%handle, size(self%surface_emissivity), self%surface_emissivity)
    hs = self%has_hard_surface
    call c_radtran_has_hard_surface_set(self%handle, hs)
    call c_radtran_photon_scale_factor_set(self%handle, self%photon_scale_factor)
    call c_radtran_ir_tau_min_set(self%handle, self%ir_tau_min)
    call c_radtran_diurnal_fac_set(self%handle, self%diurnal_fac)
  end subroutine

  subroutine pull_results(self, do_solar)
    class(Radtran), intent(inout) :: self
    logical, intent(in) :: do_solar
    type(c_ptr) :: p
    integer :: nz
    logical(c_bool) :: ds
    character(c_char) :: err_c(err_len+1)
    nz = self%nz
    call c_radtran_wrk_ir_get(self%handle, p)
    call c_climaradtranwrk_fup_n_get(p, nz+1, self%wrk_ir%fup_n)
    call c_climaradtranwrk_fdn_n_get(p, nz+1, self%wrk_ir%fdn_n)
    if (do_solar) then
      call c_radtran_wrk_sol_get(self%handle, p)
      call c_climaradtranwrk_fup_n_get(p, nz+1, self%wrk_sol%fup_n)
      call c_climaradtranwrk_fdn_n_get(p, nz+1, self%wrk_sol%fdn_n)
    endif
    if (self%sync_spectra) then
      ! the seven per-bin arrays in one go: page-locked on first use, asynchronous copies, one synchronise
      ! (the reference's holder is plain allocatables the caller reads after the call, clima_radtran.f90:11-25)
      ds = do_solar
      call c_radtran_spectra_get_all(self%handle, ds, nz+1, self%ir%nw, self%sol%nw, &
                                     self%wrk_ir%fup_a, self%wrk_ir%fdn_a, self%wrk_ir%tau_band, &
                                     self%wrk_sol%fup_a, self%wrk_sol%fdn_a, self%wrk_sol%amean, self%wrk_sol%tau_band, err_c)
    endif
    call c_radtran_f_total_get(self%handle, nz+1, self%f_total)
  end subroutine

  !> Radtran%radiate (clima_radtran.f90:221-318), same argument list
  subroutine Radtran_radiate(self, T_surface, T, P, densities, dz, pdensities, radii, compute_solar, compute_opacity, err)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: T_surface
    real(dp), intent(in) :: T(:) !! (nz) K
    real(dp), intent(in) :: P(:) !! (nz) bars
    real(dp), intent(in) :: densities(:,:) !! (nz,ng) molecules/cm3
    real(dp), intent(in) :: dz(:) !! (nz) cm
    real(dp), optional, target, intent(in) :: pdensities(:,:), radii(:,:)
    logical, optional, intent(in) :: compute_solar, compute_opacity
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    integer(c_int) :: cs, co, hp, p1, p2
    real(dp) :: dummy(1)
    logical :: do_solar

    ! check_inputs, clima_radtran.f90:426-430 (presence is only visible on this side)
    if ((present(pdensities) .and. .not. present(radii)) .or. (present(radii) .and. .not. present(pdensities))) then
      err = 'Both pdensities and radii must be arguments.'
      return
    endif
    cs = 1; co = 1
    if (present(compute_solar)) cs = merge(1, 0, compute_solar)
    if (present(compute_opacity)) co = merge(1, 0, compute_opacity)
    do_solar = cs == 1
    call push_fields(self)
    dummy = 0.0_dp
    if (present(radii)) then
      ! both shapes go across: check_dimensions_p (clima_radtran.f90:446-463) reports them separately,
      ! after the T / P / densities / dz checks
      hp = 1; p1 = size(pdensities,1); p2 = size(pdensities,2)
      call c_radtran_radiate_wrapper(self%handle, T_surface, size(T), T, size(P), P, size(densities,1), &
                                   size(densities,2), densities, size(dz), dz, hp, p1, p2, pdensities, &
                                   size(radii,1), size(radii,2), radii, cs, co, err_c)
    else
      hp = 0; p1 = 0; p2 = 0
      call c_radtran_radiate_wrapper(self%handle, T_surface, size(T), T, size(P), P, size(densities,1), &
                                   size(densities,2), densities, size(dz), dz, hp, p1, p2, dummy, p1, p2, dummy, &
                                   cs, co, err_c)
    endif
    call take_err(err_c, err)
    if (allocated(err)) return
    call pull_results(self, do_solar)
  end subroutine

  !> Radtran%TOA_fluxes (clima_radtran.f90:320-342)
  subroutine Radtran_TOA_fluxes(self, T_surface, T, P, densities, dz, pdensities, radii, &
                                compute_solar, compute_opacity, ISR, OLR, err)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: T_surface
    real(dp), intent(in) :: T(:), P(:), densities(:,:), dz(:)
    real(dp), optional, target, intent(in) :: pdensities(:,:), radii(:,:)
    logical, optional, intent(in) :: compute_solar, compute_opacity
    real(dp), intent(out) :: ISR, OLR
    character(:), allocatable, intent(out) :: err
    call self%radiate(T_surface, T, P, densities, dz, pdensities, radii, compute_solar, compute_opacity, err)
    if (allocated(err)) return
    ISR = (self%wrk_sol%fdn_n(self%nz+1) - self%wrk_sol%fup_n(self%nz+1))
    OLR = - (self%wrk_ir%fdn_n(self%nz+1) - self%wrk_ir%fup_n(self%nz+1))
  end subroutine

  subroutine Radtran_set_bolometric_flux(self, flux)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: flux !! W/m^2
    call c_radtran_set_bolometric_flux_wrapper(self%handle, flux)
    call c_radtran_photon_scale_factor_get(self%handle, self%photon_scale_factor)
  end subroutine

  function Radtran_bolometric_flux(self) result(flux)
    class(Radtran), target, intent(inout) :: self
    real(dp) :: flux
    call c_radtran_photon_scale_factor_set(self%handle, self%photon_scale_factor)
    call c_radtran_bolometric_flux_wrapper(self%handle, flux)
  end function

  function Radtran_skin_temperature(self, bond_albedo) result(T_skin)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: bond_albedo
    real(dp) :: T_skin
    call c_radtran_photon_scale_factor_set(self%handle, self%photon_scale_factor)
    call c_radtran_skin_temperature_wrapper(self%handle, bond_albedo, T_skin)
  end function

  function Radtran_equilibrium_temperature(self, bond_albedo) result(T_eq)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: bond_albedo
    real(dp) :: T_eq
    call c_radtran_photon_scale_factor_set(self%handle, self%photon_scale_factor)
    call c_radtran_equilibrium_temperature_wrapper(self%handle, bond_albedo, T_eq)
  end function

  subroutine Radtran_apply_radiation_enhancement(self, rad_enhancement)
    class(Radtran), target, intent(inout) :: self
    real(dp), intent(in) :: rad_enhancement
    call c_radtran_apply_radiation_enhancement(self%handle, rad_enhancement)
    call pull_results(self, .true.)
  end subroutine

  !> Many independent columns (the column is the last dimension of every array): each one a full
  !> `TOA_fluxes` (clima_radtran.f90:320-342), enqueued back to back on the device.
  !> `fluxes(nz+1, 5, ncol)` = ir up, ir down, solar up, solar down, f_total.
  subroutine Radtran_TOA_fluxes_batch(self, T_surface, T, P, densities, dz, pdensities, radii, ISR, OLR, fluxes, err)
    class(Radtran), intent(inout) :: self
    real(dp), intent(in) :: T_surface(:), T(:,:), P(:,:), densities(:,:,:), dz(:,:)
    real(dp), intent(in) :: pdensities(:,:,:), radii(:,:,:)
    real(dp), intent(out) :: ISR(:), OLR(:), fluxes(:,:,:)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    integer :: ncol
    ncol = size(T_surface)
    if (size(T,1) /= self%nz .or. size(T,2) /= ncol .or. any(shape(densities) /= [self%nz, self%ng, ncol]) .or. &
        any(shape(fluxes) /= [self%nz+1, 5, ncol]) .or. size(ISR) /= ncol .or. size(OLR) /= ncol) then
      err = '"T" has the wrong input dimension.'
      return
    endif
    call push_fields(self)
    call c_radtran_toa_fluxes_batch(self%handle, ncol, T_surface, T, P, densities, dz, merge(1, 0, self%np > 0), &
                                    pdensities, radii, ISR, OLR, fluxes, err_c)
    call take_err(err_c, err)
    if (allocated(err)) return
    call pull_results(self, .true.)
  end subroutine

  !> The RCE Jacobian's radiative calls in one go (src/adiabat/clima_adiabat_solve.f90:798-812):
  !> column i is `radiate(T_surface(i), T(:,i), ..., compute_solar=.false., compute_opacity=.false.)`
  !> on the opacities of the last compute_opacity call; `f_total(:,i)` is what that call would
  !> leave in `self%f_total`, `fup_n`/`fdn_n` what it would leave in `self%wrk_ir`.
  subroutine Radtran_radiate_ir_batch(self, T_surface, T, fup_n, fdn_n, f_total, err)
    class(Radtran), intent(inout) :: self
    real(dp), intent(in), contiguous :: T_surface(:)   !! (ncol)
    real(dp), intent(in), contiguous :: T(:,:)         !! (nz, ncol)
    real(dp), intent(out), contiguous :: fup_n(:,:), fdn_n(:,:), f_total(:,:)   !! (nz+1, ncol)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    integer :: ncol
    ! (`contiguous`: the arrays go to the library as they are -- the batch is ~1 ms, copies of 1-4 MB would show; a
    ! strided actual argument is copied in / out by the compiler)
    ncol = size(T_surface)
    if (size(T,2) /= ncol .or. any(shape(fup_n) /= [self%nz+1, ncol]) .or. &
        any(shape(fdn_n) /= [self%nz+1, ncol]) .or. any(shape(f_total) /= [self%nz+1, ncol])) then
      err = '"T" has the wrong input dimension.'
      return
    endif
    call c_radtran_batch_pin_results_set(self%handle, merge(1_c_int, 0_c_int, self%pin_batch_results))
    call c_radtran_radiate_ir_batch(self%handle, ncol, T_surface, size(T,1), size(T,2), T, fup_n, fdn_n, f_total, err_c)
    call take_err(err_c, err)
  end subroutine

  !> clima_radtran.f90 `opacities2yaml` (-> clima_radtran_types.f90:328-430)
  function Radtran_opacities2yaml(self) result(out)
    class(Radtran), intent(inout) :: self
    character(:), allocatable :: out
    integer(c_int) :: n
    type(c_ptr) :: cp
    character(c_char), allocatable :: buf(:)
    integer :: i
    call c_radtran_opacities2yaml_wrapper_1(self%handle, n, cp)
    allocate(buf(n+1))
    call c_radtran_opacities2yaml_wrapper_2(self%handle, cp, n, buf)
    allocate(character(n) :: out)
    do i = 1, n
      out(i:i) = buf(i)
    enddo
  end function

  !> species / particle names (index order) that opacities2yaml prints
  subroutine Radtran_set_names(self, species_names, particle_names, err)
    class(Radtran), intent(inout) :: self
    character(*), intent(in) :: species_names(:), particle_names(:)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    character(:), allocatable :: s1, s2
    integer :: i
    s1 = ''
    do i = 1, size(species_names)
      s1 = s1//trim(species_names(i))
      if (i /= size(species_names)) s1 = s1//new_line('a')
    enddo
    s2 = ''
    do i = 1, size(particle_names)
      s2 = s2//trim(particle_names(i))
      if (i /= size(particle_names)) s2 = s2//new_line('a')
    enddo
    call c_radtran_set_names(self%handle, to_c(s1), to_c(s2), err_c)
    call take_err(err_c, err)
  contains
    function to_c(s) result(c)
      character(*), intent(in) :: s
      character(c_char) :: c(len(s)+1)
      integer :: k
      do k = 1, len(s)
        c(k) = s(k:k)
      enddo
      c(len(s)+1) = c_null_char
    end function
  end subroutine

  !> clima_radtran.f90:494-506
  subroutine Radtran_set_custom_optical_properties(self, wv, P, dtau_dz, w0, g0, err)
    class(Radtran), intent(inout) :: self
    real(dp), intent(in) :: wv(:), P(:)
    real(dp), intent(in) :: dtau_dz(:,:), w0(:,:), g0(:,:)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: cerr(ERR_LEN+1)
    real(dp), allocatable :: a(:,:), b(:,:), c(:,:)
    a = dtau_dz; b = w0; c = g0   ! contiguous copies
    call c_radtran_set_custom_optical_properties(self%handle, size(wv), wv, size(P), P, &
      size(a,1), size(a,2), a, size(b,1), size(b,2), b, size(c,1), size(c,2), c, cerr)
    call take_err(cerr, err)
  end subroutine

  !> clima_radtran.f90:508-512
  subroutine Radtran_unset_custom_optical_properties(self)
    class(Radtran), intent(inout) :: self
    call c_radtran_unset_custom_optical_properties(self%handle)
  end subroutine

  subroutine Radtran_destroy(self)
    class(Radtran), intent(inout) :: self
    if (c_associated(self%handle)) then
      call c_radtran_spectra_release(self%handle)   ! the result arrays are no longer page-locked: they may be freed
      call c_deallocate_radtran(self%handle)
    endif
    self%handle = c_null_ptr
  end subroutine

  !> select this process's GPU (before `finish`)
  subroutine radtran_set_device(device, err)
    integer, intent(in) :: device
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_set_device(int(device, c_int), err_c)
    call take_err(err_c, err)
  end subroutine

  !> rank 0: a fresh communicator id, to be handed to every rank (MPI_Bcast of comm_id_bytes characters, ...)
  subroutine radtran_comm_unique_id(id, err)
    character(c_char), intent(out) :: id(comm_id_bytes)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_comm_unique_id(id, err_c)
    call take_err(err_c, err)
  end subroutine

  !> collective over the `nranks` processes (rank 0-based); restricts the handle to its share of the bins
  subroutine Radtran_comm_init(self, nranks, rank, id, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: nranks, rank
    character(c_char), intent(in) :: id(comm_id_bytes)
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    call c_radtran_comm_init_rank(self%handle, int(nranks, c_int), int(rank, c_int), id, err_c)
    call take_err(err_c, err)
  end subroutine

  !> the same with the id exchanged through a file that rank 0 creates (`path` new for every job)
  subroutine Radtran_comm_init_file(self, nranks, rank, path, err)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: nranks, rank
    character(*), intent(in) :: path
    character(:), allocatable, intent(out) :: err
    character(c_char) :: err_c(err_len+1)
    character(c_char), allocatable :: path_c(:)
    integer :: i, n
    n = len_trim(path)
    allocate(path_c(n+1))
    do i = 1, n
      path_c(i) = path(i:i)
    enddo
    path_c(n+1) = c_null_char
    call c_radtran_comm_init_file(self%handle, int(nranks, c_int), int(rank, c_int), path_c, err_c)
    call take_err(err_c, err)
  end subroutine

  subroutine Radtran_comm_destroy(self)
    class(Radtran), intent(inout) :: self
    if (c_associated(self%handle)) call c_radtran_comm_destroy(self%handle)
  end subroutine

  !> `radiate_ir_batch`'s response form (include/clima_radtran_hip.h, radtran_ir_green_set): 0 never, 1 (default) when
  !> the batch is large and its columns are one profile with a few temperatures changed each -- what
  !> AdiabatClimate_jacobian_from_base (src/adiabat/clima_adiabat_solve.f90:768-822) issues --, 2 whenever any column is.
  !> un-page-lock the caller arrays the library locked (`sync_spectra`'s seven, `pin_batch_results`' three): before they
  !> are deallocated while the object lives on (`destroy` does it too)
  subroutine Radtran_release_pinned(self)
    class(Radtran), intent(inout) :: self
    if (c_associated(self%handle)) call c_radtran_spectra_release(self%handle)
  end subroutine

  subroutine Radtran_set_ir_green(self, mode)
    class(Radtran), intent(inout) :: self
    integer, intent(in) :: mode
    if (c_associated(self%handle)) call c_radtran_ir_green_set(self%handle, int(mode, c_int))
  end subroutine

  !> batches that took the response form so far
  function Radtran_ir_green_batches(self) result(n)
    class(Radtran), intent(inout) :: self
    integer :: n
    integer(c_int) :: mode, batches
    n = 0
    if (.not. c_associated(self%handle)) return
    call c_radtran_ir_green_get(self%handle, mode, batches)
    n = batches
  end function

end module
