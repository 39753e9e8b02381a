"""Frame files and run directories of the reference (diffpiso/datamanagement.py): simulation frames are
`<field>_%06d.npz` with one array `arr_0` (velocity: staggered tensor [1,Ny+1,Nx+1,2]).  `make_tf_dataset` becomes
`make_dataset`, a plain Python iterator with the same shuffle / batch semantics (no TensorFlow input pipeline here)."""
import os
import shutil
from collections.abc import Iterable

import numpy as np


def create_base_dir(path, name):
    """datamanagement.py:11-22: first free `path + name + %06d`."""
    i = 0
    while os.path.exists(path + name + str(i).zfill(6)):
        i += 1
    target = path + name + str(i).zfill(6)
    try:
        os.mkdir(target)
    except OSError:
        print("error creating directory: " + path + name + str(i))
    else:
        print("Created directory  " + target)
    return target


def data_path_assembler(paths, field_names, characteristics, start_frame, frame_count, step_count, dt_ratio=1):
    """datamanagement.py:35-47: per start frame, the file names of step_count+1 consecutive frames of every field, plus
    the characteristic of that sequence."""
    file_list = tuple([[] for _ in range(len(field_names) + 1)])
    for p in range(len(paths)):
        for i in range(start_frame[p], start_frame[p] + frame_count[p] - step_count[p] * dt_ratio):
            for n in range(len(field_names)):
                file_list[n].append([paths[p] + field_names[n] + "_" + str(i + j * dt_ratio).zfill(6) + ".npz"
                                     for j in range(0, step_count[p] + 1)])
            if isinstance(characteristics[p], Iterable):
                file_list[-1].append(characteristics[p][i - start_frame[p]])
            else:
                file_list[-1].append(characteristics[p])
    return file_list


def load_function(*data_tuple):
    """datamanagement.py:50-57: stack the frames of every field along a new axis 1 ([1,T,...], float32)."""
    output = []
    for d in range(len(data_tuple) - 1):
        output.append(np.concatenate([np.expand_dims(np.load(f)["arr_0"].astype(np.float32), axis=1) for f in data_tuple[d]],
                                     axis=1))
    output.append(np.expand_dims(np.array(data_tuple[-1]), 0).astype(np.float32))
    return tuple(output)


def save_frame(path, field_name, frame, array):
    """The writer the reference's simulation scripts inline (np.savez(path + 'velocity_%06d' % i, tensor))."""
    np.savez(path + field_name + "_" + str(frame).zfill(6) + ".npz", np.asarray(array))


def make_dataset(list_tuple, mapping_func=load_function, batch_size=1, shuffle=True, seed=None):
    """datamanagement.py:25-32 without tf.data: yields batches (concatenated along axis 0) of mapped sequences."""
    order = np.arange(len(list_tuple[0]))
    if shuffle:
        np.random.default_rng(seed).shuffle(order)
    for b in range(0, len(order), batch_size):
        items = [mapping_func(*[lt[k] for lt in list_tuple]) for k in order[b:b + batch_size]]
        yield tuple(np.concatenate([it[f] for it in items], axis=0) for f in range(len(items[0])))


def save_source(file, path, filename):
    shutil.copy(file, path + filename)
