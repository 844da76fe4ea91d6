import numpy as np

dt_int = np.int32
dt_float = np.float32
dt_bool = np.bool_
