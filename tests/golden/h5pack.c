/* h5pack -- write HDF5 datasets with the HDF5 C library's own dataset-creation path: chunked layout,
 * shuffle + deflate filters, and float32 or float64 storage.  Used by make_datadir_c.py to produce
 * tests/golden/datadir_c/, a tiny opacity-data directory in the schema of the reference's loader
 * (src/radtran/clima_radtran_types_create.f90:734-1468) that was NOT written by clima_amd/h5lite.py --
 * so reading it back is not a round trip through one writer/reader pair.  Test infrastructure.
 *
 *   h5pack out.h5 spec.txt
 * spec.txt, one dataset per line:
 *   name  f32|f64  rank  d0 .. d(rank-1)  c0 .. c(rank-1)  deflate_level  raw_file
 * raw_file holds prod(d) float64 values in C order.
 * Build: gcc -O2 -I/opt/conda/include h5pack.c -o h5pack -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: h5pack out.h5 spec.txt\n"); return 2; }
  hid_t file = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (file < 0) { fprintf(stderr, "cannot create %s\n", argv[1]); return 1; }
  FILE *spec = fopen(argv[2], "r");
  if (!spec) { fprintf(stderr, "cannot open %s\n", argv[2]); return 1; }
  char name[256], type[8], raw[1024];
  int rank;
  while (fscanf(spec, "%255s %7s %d", name, type, &rank) == 3) {
    hsize_t dims[8], chunk[8], n = 1;
    int level;
    if (rank < 1 || rank > 8) return 1;
    for (int i = 0; i < rank; i++) { unsigned long long v; if (fscanf(spec, "%llu", &v) != 1) return 1; dims[i] = v; n *= v; }
    for (int i = 0; i < rank; i++) { unsigned long long v; if (fscanf(spec, "%llu", &v) != 1) return 1; chunk[i] = v; }
    if (fscanf(spec, "%d %1023s", &level, raw) != 2) return 1;
    double *buf = (double *)malloc(n * sizeof(double));
    FILE *rf = fopen(raw, "rb");
    if (!rf || fread(buf, sizeof(double), n, rf) != n) { fprintf(stderr, "cannot read %s\n", raw); return 1; }
    fclose(rf);
    hid_t space = H5Screate_simple(rank, dims, NULL);
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    if (level >= 0) {
      H5Pset_chunk(dcpl, rank, chunk);
      H5Pset_shuffle(dcpl);
      H5Pset_deflate(dcpl, (unsigned)level);
    }
    hid_t ftype = strcmp(type, "f32") == 0 ? H5T_IEEE_F32LE : H5T_IEEE_F64LE;
    hid_t dset = H5Dcreate2(file, name, ftype, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    if (dset < 0 || H5Dwrite(dset, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) {
      fprintf(stderr, "cannot write dataset %s\n", name);
      return 1;
    }
    H5Dclose(dset); H5Pclose(dcpl); H5Sclose(space); free(buf);
  }
  fclose(spec);
  H5Fclose(file);
  return 0;
}
