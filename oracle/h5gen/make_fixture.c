/* Test infrastructure (oracle/): writes an HDF5 file with the REAL libhdf5 in exactly the layout the reference's
 * external/dataset_tool_h5.py:104-111 produces through h5py (h5py defaults: libver earliest, contiguous layout):
 *     /shapes  int32 [N][3]        = (3, h, w) per image
 *     /images  vlen<uint8> [N]     = raw CHW bytes of image i
 * so that the dependency-free reader ssdn/datasets/h5lite.py can be pinned against a libhdf5-written file although h5py is not
 * installed.  Deterministic pixel values: byte k of image i = (i * 131 + k * 7 + (k >> 8)) & 255.
 *
 * build + run (build container only; the fixture is committed):
 *     gcc -O1 -I/opt/conda/include oracle/h5gen/make_fixture.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -o oracle/_build/make_fixture
 *     oracle/_build/make_fixture tests/golden/g_libhdf5_dataset.h5
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s out.h5\n", argv[0]); return 2; }
    enum { N = 5 };
    const int hs[N] = {9, 12, 7, 16, 33}, ws[N] = {17, 8, 7, 24, 5};
    hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return 1;
    hsize_t d2[2] = {N, 3}, d1[1] = {N};
    hid_t s2 = H5Screate_simple(2, d2, NULL), s1 = H5Screate_simple(1, d1, NULL);
    hid_t shapes = H5Dcreate2(f, "shapes", H5T_STD_I32LE, s2, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t vt = H5Tvlen_create(H5T_STD_U8LE);
    hid_t images = H5Dcreate2(f, "images", vt, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int32_t sh[N][3];
    hvl_t v[N];
    for (int i = 0; i < N; ++i) {
        sh[i][0] = 3; sh[i][1] = hs[i]; sh[i][2] = ws[i];
        size_t n = (size_t)3 * hs[i] * ws[i];
        uint8_t* p = (uint8_t*)malloc(n);
        for (size_t k = 0; k < n; ++k) p[k] = (uint8_t)((i * 131 + k * 7 + (k >> 8)) & 255);
        v[i].len = n; v[i].p = p;
    }
    /* h5py writes element by element (dset[idx] = ...): do the same, one hyperslab per image */
    for (int i = 0; i < N; ++i) {
        hsize_t st1[1] = {(hsize_t)i}, c1[1] = {1}, st2[2] = {(hsize_t)i, 0}, c2[2] = {1, 3};
        hid_t m1 = H5Screate_simple(1, c1, NULL), m2 = H5Screate_simple(2, c2, NULL);
        hid_t fs1 = H5Dget_space(images), fs2 = H5Dget_space(shapes);
        H5Sselect_hyperslab(fs1, H5S_SELECT_SET, st1, NULL, c1, NULL);
        H5Sselect_hyperslab(fs2, H5S_SELECT_SET, st2, NULL, c2, NULL);
        if (H5Dwrite(images, vt, m1, fs1, H5P_DEFAULT, &v[i]) < 0) return 1;
        if (H5Dwrite(shapes, H5T_NATIVE_INT32, m2, fs2, H5P_DEFAULT, sh[i]) < 0) return 1;
        H5Sclose(m1); H5Sclose(m2); H5Sclose(fs1); H5Sclose(fs2);
    }
    H5Dclose(images); H5Dclose(shapes); H5Tclose(vt); H5Sclose(s1); H5Sclose(s2); H5Fclose(f);
    return 0;
}
