"""``read_TFdata`` surface of the reference (LDPC_128/Ldpc_128_testing/read_TFdata.py:10-29)
on the TensorFlow-free TFRecord codec: ``data_handler(code_length, file_name, batch_size)`` returns
an object with ``as_numpy_iterator()`` yielding ``(soft_input, label, shape)`` batches."""
from .tfrecord import RecordDataset, decode_example


def parse_exmp(serial_exmp, code_length):
    ex = decode_example(serial_exmp)
    return ex["feature"], ex["label"], ex["shape"].astype("int32")


def get_dataset(fname, code_length):
    return RecordDataset(fname, code_length, batch_size=1)


def data_handler(code_length, file_name, batch_size=1):
    return RecordDataset(file_name, code_length, batch_size=batch_size)
