!> Fortran host program in the shape of the reference's tests/test_radtran.f90: build a
!> `Radtran`, call `rad%radiate` once, print `rad%wrk_sol%fdn_n(nz+1)*1e-3`, dump results.
!> The tables and the column come from a binary case file written by
!> clima_amd/fortran_case.py (the reference reads YAML/HDF5/atmosphere.txt here; those
!> loaders are outside the hot path).  Usage: radtran_driver case.bin result.txt
!>
!> Multi-GPU mode (the library's own RCCL step): start the program once per GPU with
!>     radtran_driver case.bin result.txt <rank> <nranks> <id-file> [<device>]
!> (rank 0-based; <id-file> a path that is new for this job; <device> defaults to <rank>).  Every rank builds
!> the same Radtran, joins the communicator (`rad%comm_init_file`) and then runs exactly the same calls: each
!> `rad%radiate` works on the rank's share of the spectral bins and ends with one all-reduce of the level
!> fluxes, so the level fluxes, f_total, ISR and OLR every rank writes are those of the whole spectrum.  Per-bin
!> spectra stay sharded, and the batched entry points are not available on a sharded handle: in this mode the
!> program stops after the two radiate calls.
program radtran_driver
  use iso_fortran_env, only: int32, output_unit
  use clima_radtran_hip, only: Radtran, dp, radtran_set_device
  implicit none
  type(Radtran) :: rad
  character(:), allocatable :: err
  character(1024) :: fin, fout
  integer(int32) :: nz, nsp, np, nw, nzen, nk, nxs, has_cont, npart, n_ir, n_sol
  integer(int32) :: sp_ind, ng, npr, nT, xs_type, xdim, sp1, sp2, LH2O, p_ind, nrad
  real(dp) :: albedo, T_surface, ISR, OLR
  real(dp), allocatable :: wavl(:), weights(:), log10P(:), temp(:), log10k(:,:,:,:), xs0(:), xs1(:,:)
  real(dp), allocatable :: h2o(:,:), frn(:,:), radii_ax(:), w0(:,:), qext(:,:), gt(:,:)
  real(dp), allocatable :: ir_wavl(:), sol_wavl(:), photons(:)
  real(dp), allocatable :: T(:), P(:), densities(:,:), dz(:), pdensities(:,:), radii(:,:)
  integer :: i, u
  integer :: my_rank, n_ranks, my_device
  character(1024) :: arg, id_file
  logical :: sharded

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  sharded = command_argument_count() >= 5
  if (sharded) then
    call get_command_argument(3, arg); read(arg, *) my_rank
    call get_command_argument(4, arg); read(arg, *) n_ranks
    call get_command_argument(5, id_file)
    my_device = my_rank
    if (command_argument_count() >= 6) then
      call get_command_argument(6, arg); read(arg, *) my_device
    endif
    call radtran_set_device(my_device, err); call check()
  endif
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) nz, nsp, np, nw, nzen
  read(u) albedo
  allocate(wavl(nw+1)); read(u) wavl
  call rad%begin(nz, nsp, np, wavl, err); call check()
  read(u) nk
  do i = 1, nk
    read(u) sp_ind, ng, npr, nT
    allocate(weights(ng), log10P(npr), temp(nT), log10k(ng,npr,nT,nw))
    read(u) weights; read(u) log10P; read(u) temp; read(u) log10k
    call rad%add_ktable(sp_ind, weights, log10P, temp, log10k, err); call check()
    deallocate(weights, log10P, temp, log10k)
  enddo
  read(u) nxs
  do i = 1, nxs
    read(u) xs_type, xdim, sp1, sp2, nT
    if (xdim == 0) then
      allocate(xs0(nw)); read(u) xs0
      call rad%add_xsection(xs_type, [sp1, sp2], xs_0d=xs0, err=err); call check()
      deallocate(xs0)
    else
      allocate(temp(nT), xs1(nT,nw)); read(u) temp; read(u) xs1
      call rad%add_xsection(xs_type, [sp1, sp2], temp=temp, log10_xs_1d=xs1, err=err); call check()
      deallocate(temp, xs1)
    endif
  enddo
  read(u) has_cont
  if (has_cont == 1) then
    read(u) LH2O, nT
    allocate(temp(nT), h2o(nT,nw), frn(nT,nw)); read(u) temp; read(u) h2o; read(u) frn
    call rad%set_water_continuum(LH2O, temp, h2o, frn, err); call check()
    deallocate(temp, h2o, frn)
  endif
  read(u) npart
  do i = 1, npart
    read(u) p_ind, nrad
    allocate(radii_ax(nrad), w0(nrad,nw), qext(nrad,nw), gt(nrad,nw))
    read(u) radii_ax; read(u) w0; read(u) qext; read(u) gt
    call rad%add_particle(p_ind, radii_ax, w0, qext, gt, err); call check()
    deallocate(radii_ax, w0, qext, gt)
  enddo
  read(u) n_ir; allocate(ir_wavl(n_ir)); read(u) ir_wavl
  read(u) n_sol; allocate(sol_wavl(n_sol)); read(u) sol_wavl
  call rad%set_channels(ir_wavl, sol_wavl, err); call check()
  allocate(photons(n_sol-1)); read(u) photons
  call rad%set_photons_sol(photons, err); call check()
  call rad%finish(nzen, albedo, err); call check()
  if (sharded) then
    call rad%comm_init_file(n_ranks, my_rank, trim(id_file), err); call check()
  endif

  allocate(T(nz), P(nz), densities(nz,nsp), dz(nz), pdensities(nz,np), radii(nz,np))
  read(u) T_surface; read(u) T; read(u) P; read(u) densities; read(u) dz
  if (np > 0) then
    read(u) pdensities; read(u) radii
  endif
  close(u)

  ! timing mode: `radtran_driver case.bin out.txt time [ncalls]` -- the synchronous drop-in call as a Fortran host sees it
  ! (SURVEY.md 8(d) "Metric": host arrays in, ISR / OLR out, device-synchronised), median of ncalls, with and without
  ! the per-bin spectra copied back into rad%wrk_*%fup_a ... after every call (rad%sync_spectra)
  if (command_argument_count() >= 3 .and. .not. sharded) then
    call get_command_argument(3, arg)
    if (trim(arg) == 'time') then
      call time_calls()
      call rad%destroy()
      stop
    endif
    if (trim(arg) == 'jac') then
      call time_jacobian()
      call rad%destroy()
      stop
    endif
  endif

  ! tests/test_radtran.f90:67
  if (np > 0) then
    call rad%radiate(T_surface, T, P, densities, dz, pdensities, radii, err=err)
  else
    call rad%radiate(T_surface, T, P, densities, dz, err=err)
  endif
  call check()
  print*, rad%wrk_sol%fdn_n(nz+1)*1.0e-3_dp   ! tests/test_radtran.f90:73

  ! the RCE-Jacobian call pattern (src/adiabat/clima_adiabat_solve.f90:811-812):
  ! IR only, opacities reused -- results must not change
  if (np > 0) then
    call rad%TOA_fluxes(T_surface, T, P, densities, dz, pdensities, radii, compute_solar=.false., &
                        compute_opacity=.false., ISR=ISR, OLR=OLR, err=err)
  else
    call rad%TOA_fluxes(T_surface, T, P, densities, dz, compute_solar=.false., compute_opacity=.false., &
                        ISR=ISR, OLR=OLR, err=err)
  endif
  call check()

  open(newunit=u, file=trim(fout), status='replace')
  write(u,'(2es26.17e3)') ISR, OLR
  write(u,'(es26.17e3)') rad%wrk_ir%fup_n
  write(u,'(es26.17e3)') rad%wrk_sol%fdn_n
  write(u,'(es26.17e3)') rad%f_total
  write(u,'(es26.17e3)') rad%wrk_ir%fup_a(nz+1,:)   ! tests/test_radtran.f90:77
  write(u,'(es26.17e3)') rad%wrk_sol%fup_a(nz+1,:)  ! :79
  if (sharded) then
    close(u)
    call rad%comm_destroy()
    call rad%destroy()
    stop
  endif

  ! the same pattern batched: three temperature columns (base, surface +1 K, layer 1 +1 K)
  block
    real(dp) :: Tsb(3), Tb(nz,3), bup(nz+1,3), bdn(nz+1,3), bft(nz+1,3)
    Tsb = T_surface; Tsb(2) = T_surface + 1.0_dp
    Tb(:,1) = T; Tb(:,2) = T; Tb(:,3) = T; Tb(1,3) = T(1) + 1.0_dp
    call rad%radiate_ir_batch(Tsb, Tb, bup, bdn, bft, err); call check()
    write(u,'(es26.17e3)') bft
    ! ... and in the response form (rad%set_ir_green: columns = one profile + a few changed temperatures)
    block
      real(dp) :: gup(nz+1,3), gdn(nz+1,3), gft(nz+1,3)
      call rad%set_ir_green(2)
      call rad%radiate_ir_batch(Tsb, Tb, gup, gdn, gft, err); call check()
      call rad%set_ir_green(1)
      write(u,'(es26.17e3)') maxval(abs(gft - bft)) / maxval(abs(bft)), real(rad%ir_green_batches(), dp)
    end block
  end block

  ! two independent columns in one batch (the second 1 K warmer): column 1 must reproduce the call above
  block
    real(dp) :: Tsb(2), Tb(nz,2), Pb(nz,2), db(nz,nsp,2), dzb(nz,2), ISRb(2), OLRb(2), fl(nz+1,5,2)
    real(dp), allocatable :: pdb(:,:,:), rab(:,:,:)
    allocate(pdb(nz,max(np,1),2), rab(nz,max(np,1),2))
    Tsb = [T_surface, T_surface + 1.0_dp]
    Tb(:,1) = T; Tb(:,2) = T + 1.0_dp
    Pb(:,1) = P; Pb(:,2) = P
    db(:,:,1) = densities; db(:,:,2) = densities
    dzb(:,1) = dz; dzb(:,2) = dz
    pdb = 1.0_dp; rab = 1.0e-5_dp
    if (np > 0) then
      pdb(:,:,1) = pdensities; pdb(:,:,2) = pdensities; rab(:,:,1) = radii; rab(:,:,2) = radii
    endif
    call rad%TOA_fluxes_batch(Tsb, Tb, Pb, db, dzb, pdb(:,1:np,:), rab(:,1:np,:), ISRb, OLRb, fl, err); call check()
    write(u,'(4es26.17e3)') ISRb, OLRb
  end block

  ! custom optical properties (clima_radtran.f90:494-512): a grey absorber-scatterer, then unset
  block
    real(dp) :: wv(3), Pc(3), dtau(3,3), w0c(3,3), g0c(3,3), ISR2, OLR2
    wv = [2.0e2_dp, 1.0e3_dp, 1.0e5_dp]
    Pc = [1.0e6_dp, 1.0e4_dp, 1.0e2_dp]
    dtau = 3.0e-8_dp; w0c = 0.5_dp; g0c = 0.3_dp
    call rad%set_custom_optical_properties(wv, Pc, dtau, w0c, g0c, err); call check()
    if (np > 0) then
      call rad%TOA_fluxes(T_surface, T, P, densities, dz, pdensities, radii, ISR=ISR2, OLR=OLR2, err=err)
    else
      call rad%TOA_fluxes(T_surface, T, P, densities, dz, ISR=ISR2, OLR=OLR2, err=err)
    endif
    call check()
    write(u,'(2es26.17e3)') ISR2, OLR2
    call rad%unset_custom_optical_properties()
  end block
  close(u)
  write(output_unit,'(a)') 'opacities2yaml:'
  write(output_unit,'(a)') rad%opacities2yaml()

  ! error convention: allocated err <=> failure, reference message text
  if (np > 0) then
    call rad%radiate(T_surface, T(1:nz-1), P, densities, dz, pdensities, radii, err=err)
  else
    call rad%radiate(T_surface, T(1:nz-1), P, densities, dz, err=err)
  endif
  if (.not. allocated(err)) then
    print*, 'expected a dimension error'
    stop 1
  endif
  write(output_unit,'(a)') 'expected error: '//err
  call rad%destroy()

contains
  ! `radtran_driver case.bin out.txt jac`: the RCE Jacobian's radiative work as a Fortran host issues it
  ! (src/adiabat/clima_adiabat_solve.f90:798-812): one full call, then nz+1 IR-only columns with one temperature changed
  ! each -- through rad%radiate_ir_batch (general kernel and response form) and one rad%radiate at a time
  subroutine time_jacobian()
    use iso_fortran_env, only: int64
    integer(int64) :: c0, c1, rate
    real(dp), allocatable :: Tsb(:), Tb(:,:), bup(:,:), bdn(:,:), bft(:,:), gft(:,:)
    real(dp) :: best(2), t_loop
    integer :: c, rep, mode
    if (np > 0) then
      call rad%radiate(T_surface, T, P, densities, dz, pdensities, radii, err=err)
    else
      call rad%radiate(T_surface, T, P, densities, dz, err=err)
    endif
    call check()
    allocate(Tsb(nz+1), Tb(nz,nz+1), bup(nz+1,nz+1), bdn(nz+1,nz+1), bft(nz+1,nz+1), gft(nz+1,nz+1))
    Tsb = T_surface; Tsb(1) = T_surface + 1.0_dp
    do c = 2, nz + 1
      Tb(:,c) = T; Tb(c-1,c) = T(c-1) + 1.0_dp
    enddo
    Tb(:,1) = T
    call system_clock(count_rate=rate)
    rad%sync_spectra = .false.
    rad%pin_batch_results = .true.     ! (bup, bdn, bft live until the end of this routine: released below)
    do mode = 0, 1
      call rad%set_ir_green(mode)
      best(mode+1) = huge(1.0_dp)
      do rep = 1, 4
        call system_clock(c0)
        call rad%radiate_ir_batch(Tsb, Tb, bup, bdn, bft, err)
        call system_clock(c1)
        call check()
        if (rep > 1) best(mode+1) = min(best(mode+1), 1.0e3_dp*real(c1 - c0, dp)/real(rate, dp))
      enddo
      if (mode == 0) gft = bft
    enddo
    call system_clock(c0)
    do c = 1, nz + 1
      if (np > 0) then
        call rad%radiate(Tsb(c), Tb(:,c), P, densities, dz, pdensities, radii, compute_solar=.false., compute_opacity=.false., err=err)
      else
        call rad%radiate(Tsb(c), Tb(:,c), P, densities, dz, compute_solar=.false., compute_opacity=.false., err=err)
      endif
    enddo
    call system_clock(c1)
    call check()
    t_loop = 1.0e3_dp*real(c1 - c0, dp)/real(rate, dp)
    write(output_unit,'(a,i0,a,i0,a,f7.2,a,f7.2,a,f8.2,a,es9.2,a,i0)') 'fortran host, RCE Jacobian: ', nz+1, ' IR-only columns x ', nz, &
      ' layers: rad%radiate_ir_batch ', best(2), ' ms (general kernel ', best(1), ' ms), one rad%radiate at a time ', t_loop, &
      ' ms; largest difference between the two batch forms ', maxval(abs(bft - gft))/maxval(abs(gft)), &
      ' of the maximum; batches in the response form: ', rad%ir_green_batches()
    rad%pin_batch_results = .false.
    call rad%release_pinned()
  end subroutine

  subroutine time_calls()
    use iso_fortran_env, only: int64
    integer :: ncalls, k, pass
    integer(int64) :: c0, c1, rate
    real(dp), allocatable :: us(:)
    real(dp) :: tmp
    integer :: a, b
    ncalls = 200
    if (command_argument_count() >= 4) then
      call get_command_argument(4, arg); read(arg, *) ncalls
    endif
    allocate(us(ncalls))
    call system_clock(count_rate=rate)
    do pass = 1, 2
      rad%sync_spectra = pass == 2
      do k = 1, 10 + ncalls
        call system_clock(c0)
        if (np > 0) then
          call rad%TOA_fluxes(T_surface, T, P, densities, dz, pdensities, radii, ISR=ISR, OLR=OLR, err=err)
        else
          call rad%TOA_fluxes(T_surface, T, P, densities, dz, ISR=ISR, OLR=OLR, err=err)
        endif
        call system_clock(c1)
        call check()
        if (k > 10) us(k-10) = 1.0e6_dp*real(c1 - c0, dp)/real(rate, dp)
      enddo
      do a = 2, ncalls        ! insertion sort: the median
        tmp = us(a); b = a - 1
        do while (b >= 1)
          if (us(b) <= tmp) exit
          us(b+1) = us(b); b = b - 1
        enddo
        us(b+1) = tmp
      enddo
      if (pass == 1) then
        write(output_unit,'(a,i0,a,f8.1,a,f8.1,a,f8.1,a)') 'fortran host, rad%TOA_fluxes x ', ncalls, &
          ' (level fluxes only): median ', us((ncalls+1)/2), ' us  p10 ', us(max(1,ncalls/10)), '  p90 ', us(max(1,(9*ncalls)/10)), ' us'
      else
        write(output_unit,'(a,i0,a,f8.1,a)') 'fortran host, rad%TOA_fluxes x ', ncalls, &
          ' (+ every per-bin spectrum copied back, rad%sync_spectra = .true.): median ', us((ncalls+1)/2), ' us'
      endif
    enddo
    write(output_unit,'(a,2es24.15)') 'ISR, OLR (mW/m^2): ', ISR, OLR
  end subroutine

  subroutine check()
    if (allocated(err)) then
      print*, err
      stop 1
    endif
  end subroutine
end program
