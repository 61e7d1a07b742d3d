"""dgl.data.utils.save_info / load_info: a pickled dict (graph_generator.py:898-899, 908)."""
import pickle


def save_info(path, info):
    with open(path, 'wb') as fh:
        pickle.dump(info, fh)


def load_info(path):
    with open(path, 'rb') as fh:
        return pickle.load(fh)
