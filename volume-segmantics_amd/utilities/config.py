"""Constants of the hot path (values from the reference's volume_segmantics/utilities/config.py:1-42)."""
# CLI argument names / file-type sets
TRAIN_DATA_ARG, LABEL_DATA_ARG, MODEL_PTH_ARG, PREDICT_DATA_ARG, DATA_DIR_ARG = "data", "labels", "model", "data", "data_dir"
TIFF_SUFFIXES = {".tiff", ".tif"}
HDF5_SUFFIXES = {".h5", ".hdf5", ".nxs"}
NUMPY_SUFFIXES = {".npy"}  # extension of this engine: raw arrays, no third-party I/O library needed
TRAIN_DATA_EXT = LABEL_DATA_EXT = PREDICT_DATA_EXT = HDF5_SUFFIXES | TIFF_SUFFIXES | NUMPY_SUFFIXES
MODEL_DATA_EXT = {".pytorch", ".pth"}
LOGGING_FMT = "%(asctime)s - %(levelname)s - %(message)s"
LOGGING_DATE_FMT = "%d-%b-%y %H:%M:%S"
SETTINGS_DIR = "volseg-settings"
TRAIN_SETTINGS_FN = "2d_model_train_settings.yaml"
PREDICTION_SETTINGS_FN = "2d_model_predict_settings.yaml"
TQDM_BAR_FORMAT = "{l_bar}{bar: 30}{r_bar}{bar: -30b}"
HDF5_COMPRESSION = "gzip"

# batch-size heuristic of the reference (config.py:29-32); MI355X-specific overrides live in the settings
BIG_CUDA_THRESHOLD = 8      # GB of free device memory above which the "big" batch sizes are used
BIG_CUDA_TRAIN_BATCH = 12
BIG_CUDA_PRED_BATCH = 4
SMALL_CUDA_BATCH = 2
NUM_WORKERS = 4
PIN_CUDA_MEMORY = True
IM_SIZE_DIVISOR = 32        # network input height/width must be a multiple of this
MODEL_INPUT_CHANNELS = 1

DEFAULT_MIN_LR = 0.00075    # LR finder fallback
LR_DIVISOR = 3
IMAGENET_MEAN = 0.449       # single-channel ImageNet statistics used for normalisation
IMAGENET_STD = 0.226

# engine-specific defaults (new optional settings keys; absent keys reproduce the reference)
HIP_PRED_BATCH = 32         # eval-mode results do not depend on the batch size, so fill the GPU
