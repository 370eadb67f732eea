/* Writes a Keras-2.2-layout HDF5 weight file with the REAL HDF5 library (the one h5py wraps): the structural ground
 * truth tests/test_host.py pins longterm360fov_amd/keras_h5.py against.  Layout = keras/engine/saving.py
 * save_weights_to_hdf5_group: root attributes layer_names / backend / keras_version (fixed-length byte strings, as
 * h5py stores numpy 'S' arrays), one group per layer with the attribute weight_names, one contiguous float32 dataset
 * per weight addressed by its name relative to the layer group ("lstm_1/kernel:0" -> nested group lstm_1).
 * Values: v[i] = ((37 i + 11 k) mod 1000 - 500) / 256 for the k-th tensor of the file (exact in float32).
 * Build + run (authoring container only; the .h5 files are the committed fixtures):
 *     /opt/conda/bin/h5cc -o /tmp/mk tests/golden/make_keras_h5_real.c && /tmp/mk tests/golden
 * argv[1] = output directory.  Writes keras_real_weights.h5 (model.save_weights) and keras_real_model.h5 (model.save:
 * the same tree under model_weights/), and both again with the scalar attributes as VARIABLE-LENGTH strings
 * (keras_real_weights_vlen.h5, keras_real_model_vlen.h5) - the layout h5py really produces for them. */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_tensor = 0;

static void str_array_attr(hid_t obj, const char* name, const char** vals, int n) {
    size_t w = 1;
    for (int i = 0; i < n; ++i) if (strlen(vals[i]) > w) w = strlen(vals[i]);
    char* buf = (char*)calloc((size_t)(n > 0 ? n : 1), w);
    for (int i = 0; i < n; ++i) memcpy(buf + (size_t)i * w, vals[i], strlen(vals[i]));
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, w);
    H5Tset_strpad(t, H5T_STR_NULLPAD);
    hsize_t dims[1] = {(hsize_t)n};
    hid_t s = H5Screate_simple(1, dims, NULL);
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    if (n > 0) H5Awrite(a, t, buf);
    H5Aclose(a); H5Sclose(s); H5Tclose(t); free(buf);
}

static void str_scalar_attr(hid_t obj, const char* name, const char* val) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, strlen(val));
    H5Tset_strpad(t, H5T_STR_NULLPAD);
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, val);
    H5Aclose(a); H5Sclose(s); H5Tclose(t);
}

/* what h5py does for `f.attrs['backend'] = b'tensorflow'` (a Python bytes / str scalar): a VARIABLE-LENGTH string, the
 * attribute holds a global-heap reference.  keras/engine/saving.py writes backend, keras_version, model_config and
 * training_config this way. */
static void vlen_scalar_attr(hid_t obj, const char* name, const char* val) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, &val);
    H5Aclose(a); H5Sclose(s); H5Tclose(t);
}
static int g_vlen = 0;
static void scalar_attr(hid_t obj, const char* name, const char* val) {
    if (g_vlen) vlen_scalar_attr(obj, name, val); else str_scalar_attr(obj, name, val);
}

static void dataset(hid_t group, const char* name, int rank, const hsize_t* dims) {
    size_t n = 1;
    for (int i = 0; i < rank; ++i) n *= (size_t)dims[i];
    float* v = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) v[i] = (float)((long)((37 * i + 11 * (size_t)g_tensor) % 1000) - 500) / 256.0f;
    ++g_tensor;
    hid_t s = H5Screate_simple(rank, dims, NULL);
    hid_t lc = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lc, 1);          /* "lstm_1/kernel:0" creates the nested group, as h5py does */
    hid_t d = H5Dcreate2(group, name, H5T_IEEE_F32LE, s, lc, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
    H5Dclose(d); H5Pclose(lc); H5Sclose(s); free(v);
}

static void lstm(hid_t root, const char* name, int F, int H) {
    hid_t g = H5Gcreate2(root, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char a[64], b[64], c[64];
    snprintf(a, 64, "%s/kernel:0", name); snprintf(b, 64, "%s/recurrent_kernel:0", name); snprintf(c, 64, "%s/bias:0", name);
    const char* wn[3] = {a, b, c};
    str_array_attr(g, "weight_names", wn, 3);
    hsize_t dk[2] = {(hsize_t)F, (hsize_t)(4 * H)}, dr[2] = {(hsize_t)H, (hsize_t)(4 * H)}, db[1] = {(hsize_t)(4 * H)};
    dataset(g, a, 2, dk); dataset(g, b, 2, dr); dataset(g, c, 1, db);
    H5Gclose(g);
}

static void dense(hid_t root, const char* name, int In, int Out) {
    hid_t g = H5Gcreate2(root, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char a[64], b[64];
    snprintf(a, 64, "%s/kernel:0", name); snprintf(b, 64, "%s/bias:0", name);
    const char* wn[2] = {a, b};
    str_array_attr(g, "weight_names", wn, 2);
    hsize_t dk[2] = {(hsize_t)In, (hsize_t)Out}, db[1] = {(hsize_t)Out};
    dataset(g, a, 2, dk); dataset(g, b, 1, db);
    H5Gclose(g);
}

static void input(hid_t root, const char* name) {
    hid_t g = H5Gcreate2(root, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    str_array_attr(g, "weight_names", NULL, 0);
    H5Gclose(g);
}

/* FoV_seq2seq.py:82-101 at toy size: F_enc 5, F_dec 3, latent_dim 4 */
static void weights_tree(hid_t root) {
    const char* layers[5] = {"input_1", "input_2", "lstm_1", "lstm_2", "dense_1"};
    str_array_attr(root, "layer_names", layers, 5);
    scalar_attr(root, "backend", "tensorflow");
    scalar_attr(root, "keras_version", "2.2.4");
    input(root, "input_1"); input(root, "input_2");
    lstm(root, "lstm_1", 5, 4); lstm(root, "lstm_2", 3, 4); dense(root, "dense_1", 4, 3);
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    char path[512];
    /* [0]: scalar attributes as fixed-length strings; [1]: as variable-length strings (the h5py / Keras form) */
    const char* wname[2] = {"keras_real_weights.h5", "keras_real_weights_vlen.h5"};
    const char* mname[2] = {"keras_real_model.h5", "keras_real_model_vlen.h5"};
    for (g_vlen = 0; g_vlen < 2; ++g_vlen) {
        snprintf(path, 512, "%s/%s", dir, wname[g_vlen]);
        hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        g_tensor = 0;
        weights_tree(f);
        H5Fclose(f);
        snprintf(path, 512, "%s/%s", dir, mname[g_vlen]);
        f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        scalar_attr(f, "keras_version", "2.2.4");
        scalar_attr(f, "backend", "tensorflow");
        scalar_attr(f, "model_config", "{\"class_name\": \"Model\", \"config\": {\"name\": \"model_1\"}}");
        scalar_attr(f, "training_config", "{\"optimizer_config\": {\"class_name\": \"Adam\"}, \"loss\": \"mean_squared_error\"}");
        hid_t mw = H5Gcreate2(f, "model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        g_tensor = 0;
        weights_tree(mw);
        H5Gclose(mw);
        H5Fclose(f);
    }
    return 0;
}
