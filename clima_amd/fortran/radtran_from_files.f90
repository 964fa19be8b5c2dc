!> Fortran host in the shape of the reference's tests/test_radtran.f90 that builds its Radtran from FILES, as the
!> reference does (`rad = Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`,
!> src/radtran/clima_radtran.f90:98-126) -- no Python and none of the reference's loaders in the loop: the files are
!> read behind the C ABI (radtran_create_from_files).
!>
!>   radtran_from_files <settings.yaml> <star.txt> <datadir> <nz> <num_zenith_angles> <surface_albedo> <column.bin> <out.txt>
!>
!> column.bin (stream): T_surface, T(nz), P(nz), densities(nz,ng), dz(nz) [, pdensities(nz,np), radii(nz,np)] as float64.
program radtran_from_files
  use clima_radtran_hip, only: Radtran, dp
  implicit none
  type(Radtran) :: rad
  character(:), allocatable :: err
  character(1024) :: settings_f, star_f, datadir, colfile, outfile, arg
  integer :: nz, nzen, u
  real(dp) :: albedo, T_surface, ISR, OLR
  real(dp), allocatable :: T(:), P(:), densities(:,:), dz(:), pdensities(:,:), radii(:,:)

  call get_command_argument(1, settings_f)
  call get_command_argument(2, star_f)
  call get_command_argument(3, datadir)
  call get_command_argument(4, arg); read(arg, *) nz
  call get_command_argument(5, arg); read(arg, *) nzen
  call get_command_argument(6, arg); read(arg, *) albedo
  call get_command_argument(7, colfile)
  call get_command_argument(8, outfile)

  rad = Radtran(trim(settings_f), trim(star_f), nzen, albedo, nz, trim(datadir), err)
  if (allocated(err)) then
    print '(a)', 'error: '//err
    stop 1
  endif

  allocate(T(nz), P(nz), densities(nz,rad%ng), dz(nz), pdensities(nz,rad%np), radii(nz,rad%np))
  open(newunit=u, file=trim(colfile), access='stream', form='unformatted', status='old')
  read(u) T_surface; read(u) T; read(u) P; read(u) densities; read(u) dz
  if (rad%np > 0) then
    read(u) pdensities; read(u) radii
  endif
  close(u)

  if (rad%np > 0) then
    call rad%TOA_fluxes(T_surface, T, P, densities, dz, pdensities, radii, ISR=ISR, OLR=OLR, err=err)
  else
    call rad%TOA_fluxes(T_surface, T, P, densities, dz, ISR=ISR, OLR=OLR, err=err)
  endif
  if (allocated(err)) then
    print '(a)', 'error: '//err
    stop 1
  endif
  print*, rad%wrk_sol%fdn_n(nz+1)*1.0e-3_dp   ! tests/test_radtran.f90:73

  open(newunit=u, file=trim(outfile), status='replace')
  write(u,'(2es26.17e3)') ISR, OLR
  write(u,'(es26.17e3)') rad%wrk_ir%fup_n
  write(u,'(es26.17e3)') rad%wrk_sol%fdn_n
  write(u,'(es26.17e3)') rad%f_total
  write(u,'(es26.17e3)') rad%wrk_ir%fup_a(nz+1,:)
  write(u,'(es26.17e3)') rad%wrk_sol%amean(1,:)
  write(u,'(es26.17e3)') rad%photons_sol
  close(u)
  print '(a)', 'opacities2yaml:'
  print '(a)', rad%opacities2yaml()
  call rad%destroy()
end program
