#!/usr/bin/env python3
"""FUNSD annotation JSONs -> preprocessed pickles (counterpart of the reference script of the same name,
funsd_preprocessing_word_level.py:117-126): training dir defines the charset, test dir reuses it."""
import argparse
import pickle

from msau_amd.data.funsd import get_preprocessed_list_word_msau

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-dir", default="dataset/training_data/annotations/")
    ap.add_argument("--test-dir", default="dataset/testing_data/annotations/")
    a = ap.parse_args()
    train_data_list, inv_dict_charset = get_preprocessed_list_word_msau(a.train_dir)
    test_data_list, inv_dict_charset = get_preprocessed_list_word_msau(a.test_dir, inv_dict_charset=inv_dict_charset)
    pickle.dump(inv_dict_charset, open("inv_dict_charset.pkl", "wb"))
    pickle.dump(train_data_list, open("funsd_preprocess_train_word.pkl", "wb"))
    pickle.dump(test_data_list, open("funsd_preprocess_test_word.pkl", "wb"))
