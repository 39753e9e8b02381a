"""Frame files and run directories in the reference's on-disk FORMAT (diffpiso/datamanagement.py, spatial_mixing_layer.py:60-75).

Format: a simulation is a directory of `<field>_NNNNNN.npz` files (six-digit frame number), each holding one array under
the key `arr_0` (velocity: staggered tensor [1, Ny+1, Nx+1, 2]; pressure: [1, Ny, Nx, 1]).  A training sample is a
window of `step_count + 1` frames spaced `dt_ratio` apart; loading it stacks the frames along a new axis 1
([1, T, ...], float32) and appends the window's "characteristic" (e.g. perturbation amplitudes) as a [1, ...] array.
Run directories are numbered: `<path><name>NNNNNN`, first free number.

Same function names / arguments / return layouts as the reference so that its scripts run; the TensorFlow input pipeline
(`make_tf_dataset`, `load_function_wrapper`) becomes the plain iterator `make_dataset`.
"""
import itertools
import os
import shutil

import numpy as np


def _frame_file(directory, field, frame):
    return "%s%s_%06d.npz" % (directory, field, frame)


def create_base_dir(path, name):
    """First unused `<path><name>NNNNNN` is created and returned (datamanagement.py:11-22)."""
    target = next(t for t in ("%s%s%06d" % (path, name, k) for k in itertools.count()) if not os.path.exists(t))
    try:
        os.mkdir(target)
        print("Created directory  " + target)
    except OSError as e:
        print("error creating directory: %s (%s)" % (target, e))
    return target


def data_path_assembler(paths, field_names, characteristics, start_frame, frame_count, step_count, dt_ratio=1):
    """Every window of step_count+1 frames inside [start_frame, start_frame + frame_count) of every dataset directory
    (datamanagement.py:35-47).  Returns a tuple of len(field_names) + 1 lists: per field the list of windows (each a list
    of file names), and the list of characteristics (one per window; a per-frame sequence is indexed by the window start)."""
    windows = [[] for _ in field_names]
    labels = []
    for directory, ch, first, count, steps in zip(paths, characteristics, start_frame, frame_count, step_count):
        per_frame = hasattr(ch, "__iter__")
        for start in range(first, first + count - steps * dt_ratio):
            frames = range(start, start + (steps + 1) * dt_ratio, dt_ratio)
            for out, field in zip(windows, field_names):
                out.append([_frame_file(directory, field, f) for f in frames])
            labels.append(ch[start - first] if per_frame else ch)
    return tuple(windows) + (labels,)


def load_function(*data_tuple):
    """One window: (files of field 0, files of field 1, ..., characteristic) -> ([1,T,...] float32 per field,
    [1,...] float32 characteristic) (datamanagement.py:50-57)."""
    *field_files, label = data_tuple
    fields = [np.stack([np.load(f)["arr_0"] for f in files], axis=1).astype(np.float32) for files in field_files]
    return tuple(fields) + (np.asarray(label, dtype=np.float32)[None],)


def save_frame(path, field_name, frame, array):
    """The writer the reference's simulation scripts inline (np.savez(path + 'velocity_' + str(i).zfill(6), tensor))."""
    np.savez(_frame_file(path, field_name, frame), np.asarray(array))


def load_frame(path, field_name, frame):
    """One frame as float32 (the reader the reference's training loop inlines, combined_training_integrated.py:268-269)."""
    return np.load(_frame_file(path, field_name, frame))["arr_0"].astype(np.float32)


def make_dataset(list_tuple, mapping_func=load_function, batch_size=1, shuffle=True, seed=None):
    """Stand-in for make_tf_dataset (datamanagement.py:25-32): yields batches (concatenated along axis 0) of mapped windows."""
    order = np.arange(len(list_tuple[0]))
    if shuffle:
        np.random.default_rng(seed).shuffle(order)
    for b in range(0, len(order), batch_size):
        items = [mapping_func(*[lt[k] for lt in list_tuple]) for k in order[b:b + batch_size]]
        yield tuple(np.concatenate([it[f] for it in items], axis=0) for f in range(len(items[0])))


def save_source(file, path, filename):
    shutil.copy(file, path + filename)
    print("Sourcefile saved to " + path + filename)
